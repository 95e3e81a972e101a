"""Tensor-level wrappers over the C ABI (include/lcaonet_hip.h).  PyTorch is plumbing here: it owns the
device buffers and the stream; every number is produced by liblcaonet_hip.so.  No CPU path exists."""
from __future__ import annotations

import ctypes

import torch
from torch import Tensor

from . import _lib

ACT_NONE, ACT_SILU, ACT_SIGMOID = 0, 1, 2
CUTOFF_KINDS = {"envelope": 0, "polynomial": 1, "cosine": 2, "none": 3}
RBF_KINDS = {"hydrogen": 0, "bessel": 1}
ORB_STRIDE = 12
FUSED_WIDTHS = (32, 64, 128)
FUSED_COEFF_SHAPES = tuple((c, v) for c in (32, 64, 128) for v in (32, 64, 128))   # (C, V) built into lcao_coeff_mlp_contract


def _p(t: Tensor | None):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: Tensor, name: str = "tensor") -> None:
    if not t.is_cuda:
        raise _lib.LcaoHipError(f"{name} is on {t.device}: lcaonet_amd runs on a ROCm GPU only (no CPU fallback)")


def _f32(t: Tensor, name: str = "tensor") -> Tensor:
    _dev(t, name)
    if t.dtype != torch.float32:
        raise TypeError(f"{name} must be float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _i64(t: Tensor, name: str = "index") -> Tensor:
    _dev(t, name)
    if t.dtype != torch.int64:
        t = t.to(torch.int64)
    return t if t.is_contiguous() else t.contiguous()


def _i32(t: Tensor, name: str = "index") -> Tensor:
    _dev(t, name)
    if t.dtype != torch.int32:
        raise TypeError(f"{name} must be int32")
    return t if t.is_contiguous() else t.contiguous()


# --- optional per-kernel timing (bench.py): HIP events on the launch stream around every C-ABI call ------
_PROFILE: dict | None = None


_PROFILE_ONLY: frozenset | None = None


class profile:
    """with ops.profile() as prof: ...  -> prof.summary() = {entry point: (calls, total ms)} after a sync.
    `only`: bracket just these entry points with HIP events (an event pair per call costs ~15 us of stream time,
    ~1 ms per forward when every call is bracketed)."""

    def __init__(self, only=None):
        self.only = None if only is None else frozenset(only)

    def __enter__(self):
        global _PROFILE, _PROFILE_ONLY
        self.events = {}
        _PROFILE, _PROFILE_ONLY = self.events, self.only
        return self

    def __exit__(self, *exc):
        global _PROFILE, _PROFILE_ONLY
        _PROFILE = _PROFILE_ONLY = None

    def summary(self) -> dict:
        torch.cuda.synchronize()
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v)) for k, v in self.events.items()}


def _call(name: str, *args) -> None:
    fn = getattr(_lib.load(), name)
    if _PROFILE is None or (_PROFILE_ONLY is not None and name not in _PROFILE_ONLY):
        _lib.check(fn(*args), name)
        return
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(fn(*args), name)
    b.record()
    _PROFILE.setdefault(name, []).append((a, b))


# --- sync-free index validation ----------------------------------------------------------------------------------
# The kernels index tables by atomic number and node id; the reference raises IndexError for a bad one.  Here the first
# kernel that touches an index range-checks it, CLAMPS it (no later kernel can read out of bounds) and sets a bit in a
# device flag.  The flag travels to pinned host memory with an asynchronous copy at the end of a forward and is turned
# into an exception as soon as that copy has landed (next forward, or ops.raise_pending_errors(sync=True)).
FLAG_EDGE_INDEX, FLAG_ATOMIC_NUMBER = 1, 2
_FLAG_TEXT = {FLAG_EDGE_INDEX: "edge_index holds a node id outside [0, N)",
              FLAG_ATOMIC_NUMBER: "z holds an atomic number outside [1, max_z]"}


class _ErrState:
    def __init__(self, dev):
        self.flag = torch.zeros(1, device=dev, dtype=torch.int32)
        self.host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.event = torch.cuda.Event()
        self.pending = False


_ERR: dict = {}


def _err_state(dev) -> _ErrState:
    key = torch.device(dev).index or 0
    if key not in _ERR:
        _ERR[key] = _ErrState(dev)
    return _ERR[key]


def publish_errors(dev) -> None:
    """Enqueue the device flag's copy to the host (no synchronisation)."""
    if torch.cuda.is_current_stream_capturing():
        return
    st = _err_state(dev)
    st.host.copy_(st.flag, non_blocking=True)
    st.event.record()
    st.pending = True


def raise_pending_errors(dev=None, sync: bool = False) -> None:
    """Raise LcaoHipError if a forward that has FINISHED flagged an invalid index (sync=True: wait for the last one)."""
    if torch.cuda.is_current_stream_capturing():     # event queries are not allowed while a HIP graph is being captured
        return
    for st in ([_err_state(dev)] if dev is not None else list(_ERR.values())):
        if not st.pending or not (sync or st.event.query()):
            continue
        if sync:
            st.event.synchronize()
        code, st.pending = int(st.host[0]), False
        if code:
            st.flag.zero_()
            st.host.zero_()
            what = "; ".join(t for b, t in _FLAG_TEXT.items() if code & b)
            raise _lib.LcaoHipError(f"invalid index in an earlier forward: {what} (the offending values were clamped on "
                                    "the device, so that call's result is meaningless)")


def checked_index(v: Tensor, lo: int, hi: int, bit: int) -> Tensor:
    """v clamped to [lo, hi] (int64); values outside set `bit` of the device error flag."""
    v = _i64(v, "index")
    out = torch.empty_like(v)
    _call("lcao_checked_index", _p(v), v.numel(), lo, hi, bit, _p(out), _p(_err_state(v.device).flag), _stream())
    return out


# lcao_build_csr (two rocPRIM radix sorts + lower bounds) is capturable into a HIP graph: profiles/r02/csr_capture_probe.txt
def build_csr(idx_s: Tensor, idx_t: Tensor, n_nodes: int) -> dict:
    """All int32 index structures of one forward in one C-ABI call (no host sync, no torch sort; capture-safe)."""
    idx_s, idx_t = _i64(idx_s, "edge_index[0]"), _i64(idx_t, "edge_index[1]")
    E, dev = idx_s.numel(), idx_s.device
    i32 = lambda n: torch.empty(n, device=dev, dtype=torch.int32)
    out = dict(order=i32(E), src_p=i32(E), dst_p=i32(E), out_ptr=i32(n_nodes + 1), in_ptr=i32(n_nodes + 1),
               in_pos=i32(E), src_o=i32(E), dst_o=i32(E))
    lib = _lib.load()
    ws_bytes = int(lib.lcao_build_csr_workspace_bytes(E, n_nodes))
    ws = torch.empty(ws_bytes, device=dev, dtype=torch.uint8)
    _call("lcao_build_csr", _p(idx_s), _p(idx_t), E, n_nodes, _p(out["order"]), _p(out["src_p"]), _p(out["dst_p"]),
          _p(out["out_ptr"]), _p(out["in_ptr"]), _p(out["in_pos"]), _p(out["src_o"]), _p(out["dst_o"]), _p(ws), ws_bytes,
          _p(_err_state(dev).flag), _stream())
    return out


def edge_geometry(pos, idx_s, idx_t, shift, lattice, batch):
    pos, shift, lattice = _f32(pos, "pos"), _f32(shift, "edge_shift"), _f32(lattice, "lattice")
    idx_s, idx_t = _i64(idx_s), _i64(idx_t)
    batch = _i64(batch) if batch is not None else None
    E = idx_s.numel()
    dist = torch.empty(E, device=pos.device, dtype=torch.float32)
    uvec = torch.empty(E, 3, device=pos.device, dtype=torch.float32)
    _call("lcao_edge_geometry", _p(pos), _p(idx_s), _p(idx_t), _p(shift), _p(lattice), _p(batch), E, _p(dist),
                                      _p(uvec), _stream())
    return dist, uvec


def cutoff(r: Tensor, kind: str, rc: float) -> Tensor:
    r = _f32(r, "r")
    out = torch.empty_like(r)
    _call("lcao_cutoff", _p(r), r.numel(), CUTOFF_KINDS[kind], float(rc), _p(out), _stream())
    return out


def radial_basis(r: Tensor, orb_table: Tensor, rbf_kind: str, cutoff_kind: str, rc: float) -> Tensor:
    r = _f32(r, "r")
    orb_table = _f32(orb_table, "orb_table")
    D = orb_table.shape[0]
    rb = torch.empty(r.numel(), D, device=r.device, dtype=torch.float32)
    _call("lcao_radial_basis", _p(r), r.numel(), D, _p(orb_table), RBF_KINDS[rbf_kind],
                                     CUTOFF_KINDS[cutoff_kind], float(rc), _p(rb), _stream())
    return rb


def edge_geometry_basis(pos, order, src_p, dst_p, shift, lattice, batch, orb_table, rbf_kind: str, cutoff_kind: str, rc: float):
    """(dist (E), uvec (E,3), uvec_p (E,3), rb (E,D)): edge_geometry + radial_basis in one launch, plus the unit vectors
    in p-space (csrc/lcao_generic.hip::k_edge_geometry_basis)."""
    pos, shift, lattice = _f32(pos, "pos"), _f32(shift, "edge_shift"), _f32(lattice, "lattice")
    orb_table = _f32(orb_table, "orb_table")
    batch = _i64(batch) if batch is not None else None
    E, D, dev = order.numel(), orb_table.shape[0], pos.device
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    dist, uvec, uvec_p, rb = new(E), new(E, 3), new(E, 3), new(E, D)
    _call("lcao_edge_geometry_basis", _p(pos), _p(_i32(order)), _p(_i32(src_p)), _p(_i32(dst_p)), _p(shift), _p(lattice),
          _p(batch), E, D, _p(orb_table), RBF_KINDS[rbf_kind], CUTOFF_KINDS[cutoff_kind], float(rc), _p(dist), _p(uvec),
          _p(uvec_p), _p(rb), _stream())
    return dist, uvec, uvec_p, rb


def sph_basis(costheta: Tensor, l_of_d: Tensor) -> Tensor:
    c = _f32(costheta, "costheta")
    l_of_d = _i32(l_of_d, "l_of_d")
    D = l_of_d.numel()
    out = torch.empty(c.numel(), D, device=c.device, dtype=torch.float32)
    _call("lcao_sph_basis", _p(c), c.numel(), D, _p(l_of_d), _p(out), _stream())
    return out


def triplet_angles(uvec: Tensor, e_st: Tensor, e_ks: Tensor) -> Tensor:
    uvec, e_st, e_ks = _f32(uvec), _i64(e_st), _i64(e_ks)
    out = torch.empty(e_st.numel(), device=uvec.device, dtype=torch.float32)
    _call("lcao_triplet_angles", _p(uvec), _p(e_st), _p(e_ks), e_st.numel(), _p(out), _stream())
    return out


def dense(x: Tensor, weight: Tensor, bias: Tensor | None = None, act: int = ACT_NONE,
          residual: Tensor | None = None, out: Tensor | None = None) -> Tensor:
    """act(x @ weight.T + bias) + residual on the last dim of x (any leading dims).  out: (M, N) contiguous rows to
    write into (e.g. a row range of a larger table)."""
    x = _f32(x, "x")
    weight = _f32(weight, "weight")
    N, K = weight.shape
    if x.shape[-1] != K:
        raise ValueError(f"dense: x has {x.shape[-1]} features, weight expects {K}")
    lead = x.shape[:-1]
    x2 = x.reshape(-1, K)
    M = x2.shape[0]
    bias = _f32(bias, "bias") if bias is not None else None
    if out is not None:
        if out.shape != (M, N) or out.dtype != torch.float32 or not out.is_contiguous():
            raise ValueError("dense: out must be a contiguous float32 (M, N) block")
        y = out
    else:
        y = torch.empty(M, N, device=x.device, dtype=torch.float32)
    r2 = None
    if residual is not None:
        r2 = _f32(residual, "residual").reshape(M, N)
    _call("lcao_dense", _p(x2), K, _p(weight), _p(bias), _p(r2), N, _p(y), N, M, K, N, act, _stream())
    return y.reshape(*lead, N)


def node_tables(x: Tensor, w_cat: Tensor, b_cat: Tensor, V: int, out=None, want_ps: bool = True):
    """(gate, Ps, Pt), each (M, V), from x (M, H) in one launch: [gate | Ps | Pt] = x w_cat^T + b_cat with sigmoid on the
    gate columns (csrc/lcao_node.hip).  out: three preallocated (M, V) blocks (row ranges of larger tables)."""
    x, w_cat, b_cat = _f32(x, "x"), _f32(w_cat, "w_cat"), _f32(b_cat, "b_cat")
    M, H = x.shape
    if w_cat.shape != (3 * V, H) or b_cat.numel() != 3 * V:
        raise ValueError("node_tables: w_cat must be (3V, H) and b_cat (3V)")
    if out is None:
        new = lambda: torch.empty(M, V, device=x.device, dtype=torch.float32)
        out = (new(), new() if want_ps else None, new())
    gate, Ps, Pt = out
    for t in out:
        if t is not None and (t.shape != (M, V) or not t.is_contiguous()):
            raise ValueError("node_tables: outputs must be contiguous (M, V) blocks")
    _call("lcao_node_tables", _p(x), _p(w_cat), _p(b_cat), M, H, V, _p(gate), _p(Ps), _p(Pt), _stream())
    return gate, Ps, Pt


def segment_sum(a: Tensor, ptr: Tensor, b: Tensor | None = None) -> Tensor:
    a = _f32(a, "a")
    ptr = _i32(ptr, "ptr")
    n_seg = ptr.numel() - 1
    width = a.shape[1] if a.dim() > 1 else 1
    b = _f32(b, "b") if b is not None else None
    if a.shape[0] == 0:   # no rows at all (edgeless graph): every segment is empty
        return torch.zeros((n_seg, width), device=a.device, dtype=torch.float32)
    out = torch.empty((n_seg, width), device=a.device, dtype=torch.float32)
    _call("lcao_segment_sum", _p(a), _p(b), _p(ptr), n_seg, width, _p(out), _stream())
    return out


def segment_sum_gather(a: Tensor, idx: Tensor, ptr: Tensor) -> Tensor:
    """out[s] = sum of a[idx[r]] over r in [ptr[s], ptr[s+1]) — gather_rows + segment_sum in one pass."""
    a, idx, ptr = _f32(a, "a"), _i32(idx, "idx"), _i32(ptr, "ptr")
    n_seg = ptr.numel() - 1
    width = a.shape[1] if a.dim() > 1 else 1
    if a.shape[0] == 0 or idx.numel() == 0:
        return torch.zeros((n_seg, width), device=a.device, dtype=torch.float32)
    out = torch.empty((n_seg, width), device=a.device, dtype=torch.float32)
    _call("lcao_segment_sum_gather", _p(a), _p(idx), _p(ptr), n_seg, width, _p(out), _stream())
    return out


def gather_rows(table: Tensor, idx: Tensor) -> Tensor:
    table = _f32(table, "table")
    width = int(table[0].numel()) if table.shape[0] > 0 else int(torch.tensor(table.shape[1:]).prod())
    out = torch.empty((idx.numel(),) + tuple(table.shape[1:]), device=table.device, dtype=torch.float32)
    if idx.dtype == torch.int32:
        idx = _i32(idx)
        _call("lcao_permute_rows", _p(table), _p(idx), idx.numel(), width, _p(out), _stream())
    else:
        idx = _i64(idx)
        _call("lcao_gather_rows", _p(table), _p(idx), idx.numel(), width, _p(out), _stream())
    return out


def _mask_args(mask: Tensor | None):
    if mask is None:
        return None, 0, 0, 0
    _dev(mask, "valence_mask")
    if mask.dtype != torch.float32:
        mask = mask.float()
    if mask.dim() == 2:
        return mask, mask.stride(0), mask.stride(1), 0
    return mask, mask.stride(0), mask.stride(1), mask.stride(2)


def coeff_contract(cstf, rb, order, l_of_d, mask, V, n_l, valence):
    cstf, rb = _f32(cstf, "cstf"), _f32(rb, "rb")
    E, D = rb.shape
    mask, mse, msd, msv = _mask_args(mask if valence else None)
    A = torch.empty(E, n_l, V, device=rb.device, dtype=torch.float32)
    B = torch.empty(E, V, device=rb.device, dtype=torch.float32)
    Bv = torch.empty(E, V, device=rb.device, dtype=torch.float32) if valence else None
    _call("lcao_coeff_contract", _p(cstf), _p(rb), _p(order), _p(l_of_d), _p(mask), mse, msd, msv, E, D, V, n_l,
                                       int(valence), _p(A), _p(B), _p(Bv), _stream())
    return A, B, Bv


LOWP = ("bf16x3", "f16")    # precision modes of the 16-bit matrix path (csrc/lcao_lowp.hip)
LOWP_MAX_ORBITALS = 32      # the 16-bit coefficient kernel keeps the orbitals of each l in a 32-bit mask; more orbitals
                            # (e.g. max_z = 96 with n_per_orb = 2) run the fp32 coefficient kernel, the rest of the layer in 16 bits


_PREC_CODE = {"bf16x3": 0, "f16": 1}    # LCAO_PREC_* of include/lcaonet_hip.h


def lowp_pack_rows(rows: Tensor, precision: str) -> Tensor:
    """MFMA operands of the 16-bit coefficient kernels (lcao_lowp_pack_rows): per 8 values 16 bytes of fp16 ("f16") or 8
    bf16 high + 8 bf16 low terms ("bf16x3").  `rows`: (..., C) fp32 with C % 32 == 0; returns a flat uint8 tensor."""
    rows = _f32(rows, "rows")
    if rows.shape[-1] % 32:
        raise ValueError(f"lowp_pack_rows: rows of {rows.shape[-1]} values (need a multiple of 32)")
    code = _PREC_CODE[precision]
    n = rows.numel()
    out = torch.empty(_lib.load().lcao_lowp_packed_bytes(n, code), device=rows.device, dtype=torch.uint8)
    _call("lcao_lowp_pack_rows", _p(rows), n, code, _p(out), _stream())
    return out


def coeff_mlp_contract_bf16x3(cst, rb, order, l_of_d, W1, W2, n_l, gate=None, src_p=None, cst_row=None, precision="bf16x3",
                              cst16=None):
    """16-bit-MFMA variant of coeff_mlp_contract (precision "bf16x3": split operands, ~1e-5; "f16": one fp16 MFMA per
    product, ~3e-4); same outputs (no valence).  cst16: `lowp_pack_rows(cst, precision)` when the caller has it (cst is
    the same tensor in every interaction, so LCAONet packs it once per forward); otherwise it is packed here."""
    rb, W1, W2 = _f32(rb, "rb"), _f32(W1), _f32(W2)
    E, D = rb.shape
    V, C = W1.shape
    dev = rb.device
    if cst16 is None:
        cst16 = lowp_pack_rows(cst, precision)
    B = torch.empty(E, V, device=dev, dtype=torch.float32)
    gated = gate is not None
    A = None if gated else torch.empty(E, n_l, V, device=dev, dtype=torch.float32)
    Ag = torch.empty(E, n_l, V, device=dev, dtype=torch.float32) if gated else None
    gram = torch.empty(E, 10, device=dev, dtype=torch.float32) if gated else None
    if gated:
        gate = _f32(gate, "gate")
    ws = tile_queue_workspace(dev)
    _call("lcao_coeff_mlp_contract_" + precision, _p(cst16), _p(rb), _p(order), _p(cst_row), _p(l_of_d), _p(W1), _p(W2), _p(gate), _p(src_p), E,
          D, C, V, n_l, _p(A), _p(B), _p(Ag), _p(gram), _p(ws), 16, _stream())
    return (Ag if gated else A), gram, B, None


def coeff_mlp_contract(cst, rb, order, l_of_d, valence_ed, W1, W2, n_l, valence, gate=None, src_p=None, p_range=None,
                       out=None, cst_row=None):
    """Fused f_coeffs + radial contraction.  With `gate` (N,V) and `src_p`: returns (Ag, gram, B, Bv) for
    triplet_gemm; without: (A, None, B, Bv) for the exact-norm triplet_sum.
    p_range = (p0, p1): only the edges [p0, p1) of p-space, written into rows [p0, p1) of `out` = the tuple a full call
    returned or `coeff_mlp_buffers(...)` allocated (sharded graphs: the owned-centre edges run while the halo exchange
    is in flight, the ghost-centred ones after it).
    cst_row (int32, p-space): edge p reads row cst_row[p] of `cst` — the gather of the coefficient rows fused into the
    kernel (cst = a table shared by many edges, e.g. the species-pair table; nothing of size (E, D, C) exists)."""
    cst, rb, W1, W2 = _f32(cst, "cst"), _f32(rb, "rb"), _f32(W1), _f32(W2)
    E, D = rb.shape
    V, C = W1.shape
    dev = rb.device
    valence_ed = _f32(valence_ed) if valence else None
    gated = gate is not None
    if out is None:
        out = coeff_mlp_buffers(E, V, n_l, dev, gated, valence)
    A, Ag, gram, B, Bv = out
    if gated:
        gate = _f32(gate, "gate")
    p0, p1 = (0, E) if p_range is None else p_range
    if p1 > p0:
        sl = lambda t: None if t is None else t[p0:p1]
        if order is None and p0 != 0:
            raise ValueError("coeff_mlp_contract: a p-range needs the order array")
        ws = tile_queue_workspace(dev)
        _call("lcao_coeff_mlp_contract", _p(cst), _p(rb), _p(sl(order)), _p(sl(cst_row)), _p(l_of_d), _p(valence_ed), _p(W1), _p(W2),
              _p(gate), _p(sl(src_p)), p1 - p0, D, C, V, n_l, int(valence), _p(sl(A)), _p(sl(B)), _p(sl(Bv)), _p(sl(Ag)),
              _p(sl(gram)), _p(ws), 16, _stream())
    return (Ag if gated else A), gram, B, Bv


def pair_rows(z: Tensor, src_p: Tensor, dst_p: Tensor, zdim: int) -> Tensor:
    """int32 (E): row of the (zdim^2, D, C) species-pair table each p-space edge reads (cst_row of coeff_mlp_contract)."""
    z = _i64(z, "z")
    E = src_p.numel()
    row = torch.empty(E, device=z.device, dtype=torch.int32)
    _call("lcao_pair_rows", _p(z), _p(_i32(src_p)), _p(_i32(dst_p)), E, zdim, _p(row), _stream())
    return row


_TILE_WS: dict = {}
_TILE_WS_CAPTURED: list = []      # captures made outside CapturedForward (a bare torch.cuda.graph): kept for the process
_TILE_WS_SINK: list | None = None


class capture_workspaces:
    """with ops.capture_workspaces(lst): the tile-queue workspaces allocated while a HIP graph is being captured inside this
    block are appended to `lst` — the capturing object keeps them alive exactly as long as its graph (CapturedForward)."""

    def __init__(self, sink: list):
        self.sink = sink

    def __enter__(self):
        global _TILE_WS_SINK
        self.prev, _TILE_WS_SINK = _TILE_WS_SINK, self.sink
        return self.sink

    def __exit__(self, *exc):
        global _TILE_WS_SINK
        _TILE_WS_SINK = self.prev


def tile_queue_workspace(dev) -> Tensor:
    """The 16-byte tile queue of the persistent coefficient kernels (include/lcaonet_hip.h): zero on first use, left zero
    by every launch.  One per (device, stream), kept for the life of the process — launches on one stream are ordered, and
    an address that is never freed can be baked into a captured HIP graph.  While a graph is being captured a fresh
    workspace is taken from the capture's pool and kept alive by the capturing object (capture_workspaces), so replays of
    different captures never share one and a destroyed capture releases its words."""
    dev = torch.device(dev)
    if torch.cuda.is_current_stream_capturing():
        ws = torch.zeros(4, device=dev, dtype=torch.int32)
        (_TILE_WS_SINK if _TILE_WS_SINK is not None else _TILE_WS_CAPTURED).append(ws)
        return ws
    key = (dev.index or 0, torch.cuda.current_stream(dev).cuda_stream)
    ws = _TILE_WS.get(key)
    if ws is None:
        ws = _TILE_WS[key] = torch.zeros(4, device=dev, dtype=torch.int32)
    return ws


def coeff_mlp_buffers(E, V, n_l, dev, gated: bool, valence: bool):
    """(A, Ag, gram, B, Bv) output buffers of coeff_mlp_contract (None where the mode does not produce one)."""
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    return (new(E, n_l, V) if (valence or not gated) else None, new(E, n_l, V) if gated else None,
            new(E, 10) if gated else None, new(E, V), new(E, V) if valence else None)


def pair_contract(Tf, rb, order, l_of_d, z, src_p, dst_p, gate, n_l, zdim, visit=None):
    """Tf ((zdim*zdim), D, V) pair table -> (Ag, gram, B) like coeff_mlp_contract(gate=...).
    visit: optional int32 permutation of p-space giving the order in which edges are walked (sorted by species pair)."""
    Tf, rb, gate = _f32(Tf, "Tf"), _f32(rb, "rb"), _f32(gate, "gate")
    z = _i64(z)
    E, D = rb.shape
    V = Tf.shape[-1]
    dev = rb.device
    Ag = torch.empty(E, n_l, V, device=dev, dtype=torch.float32)
    B = torch.empty(E, V, device=dev, dtype=torch.float32)
    gram = torch.empty(E, 10, device=dev, dtype=torch.float32)
    if visit is not None and V in FUSED_WIDTHS:
        _call("lcao_pair_contract_visit", _p(Tf), _p(rb), _p(order), _p(l_of_d), _p(z), _p(src_p), _p(dst_p), _p(gate),
              _p(visit), E, D, V, n_l, zdim, _p(Ag), _p(B), _p(gram), _stream())
        return Ag, gram, B
    _call("lcao_pair_contract", _p(Tf), _p(rb), _p(order), _p(l_of_d), _p(z), _p(src_p), _p(dst_p), _p(gate), E, D, V,
          n_l, zdim, _p(Ag), _p(B), _p(gram), _stream())
    return Ag, gram, B


def triplet_gemm(Ag, gram, uvec_p, in_ptr, in_pos, out_ptr, n_edges=None):
    """Centre nodes = the out_ptr.numel()-1 first nodes; writes tbw rows [0, out_ptr[-1]) (= n_edges when given)."""
    Ag, gram, uvec_p = _f32(Ag), _f32(gram), _f32(uvec_p)
    E, n_l, V = Ag.shape
    N = out_ptr.numel() - 1
    E = E if n_edges is None else n_edges
    tbw = torch.empty(E, V, device=Ag.device, dtype=torch.float32)
    _call("lcao_triplet_gemm", _p(Ag), _p(gram), _p(uvec_p), _p(in_ptr), _p(in_pos), _p(out_ptr), N, E, V, n_l, _p(tbw),
          _stream())
    return tbw


def triplet_sum(A, gate, uvec, order, src_p, in_ptr, in_pos, n_edges=None):
    """n_edges: compute only the first n_edges out-edges (sharded graphs: the owned prefix of p-space)."""
    A, gate, uvec = _f32(A), _f32(gate), _f32(uvec)
    E, n_l, V = A.shape
    E = E if n_edges is None else n_edges
    tbw = torch.empty(E, V, device=A.device, dtype=torch.float32)
    _call("lcao_triplet_sum", _p(A), _p(gate), _p(uvec), _p(order), _p(src_p), _p(in_ptr), _p(in_pos), E, V, n_l,
                                    _p(tbw), _stream())
    return tbw


def edge_combine(ft, B, Bv):
    ft, B = _f32(ft), _f32(B)
    E, V = B.shape
    lw = torch.empty(E, V, device=B.device, dtype=torch.float32)
    _call("lcao_edge_combine", _p(ft), _p(B), _p(Bv), E, V, _p(lw), _stream())
    return lw


def pair_silu(Ps, Pt, src_p, dst_p):
    Ps, Pt = _f32(Ps), _f32(Pt)
    E, V = src_p.numel(), Ps.shape[1]
    h1 = torch.empty(E, V, device=Ps.device, dtype=torch.float32)
    _call("lcao_pair_silu", _p(Ps), _p(Pt), _p(src_p), _p(dst_p), E, V, _p(h1), _stream())
    return h1


def embed_coeffs(FE, U, Vt, scale, shift, z, src_p, dst_p):
    FE, U, Vt, scale, shift = _f32(FE), _f32(U), _f32(Vt), _f32(scale), _f32(shift)
    z = _i64(z)
    _, D, C = FE.shape
    E = src_p.numel()
    cst = torch.empty(E, D, C, device=FE.device, dtype=torch.float32)
    _call("lcao_embed_coeffs", _p(FE), _p(U), _p(Vt), _p(scale), _p(shift), _p(z), _p(src_p), _p(dst_p), E, D, C,
                                     _p(cst), _stream())
    return cst


def triplet_literal(cstf, rb, shb, xk, tri_k, e_ks, mask, V, valence):
    cstf, rb, shb = _f32(cstf), _f32(rb), _f32(shb)
    _dev(xk)
    if xk.stride(-1) != 1:
        xk = xk.contiguous()
    T, D = shb.shape
    tri_k, e_ks = _i64(tri_k), _i64(e_ks)
    mask, mse, msd, msv = _mask_args(mask if valence else None)
    w = torch.empty(T, V, device=rb.device, dtype=torch.float32)
    _call("lcao_triplet_literal", _p(cstf), _p(rb), _p(shb), _p(xk), xk.stride(0), _p(tri_k), _p(e_ks), _p(mask),
                                        mse, msd, msv, T, D, V, int(valence), _p(w), _stream())
    return w


def edge_lcao_weight(tbw, B, W3, Wb, precision="fp32", save=False):
    """bw, or (bw, ft) with save=True (fp32 only): ft = tbw W3^T is what the reverse sweep needs."""
    tbw, B, W3, Wb = _f32(tbw), _f32(B), _f32(W3), _f32(Wb)
    E, V = B.shape
    bw = torch.empty(E, V, device=B.device, dtype=torch.float32)
    if save:
        ft = torch.empty(E, V, device=B.device, dtype=torch.float32)
        _call("lcao_edge_lcao_weight_saving", _p(tbw), _p(B), _p(W3), _p(Wb), E, V, _p(bw), _p(ft), _stream())
        return bw, ft
    _call("lcao_edge_lcao_weight_" + precision if precision in LOWP else "lcao_edge_lcao_weight", _p(tbw), _p(B), _p(W3), _p(Wb), E, V,
          _p(bw), _stream())
    return bw


def edge_node_message(bw, Ps, Pt, src_p, dst_p, Wn2, bn2, precision="fp32", save=False):
    """msg, or (msg, pre2) with save=True (fp32 only): pre2 is the pre-activation of the second f_node layer."""
    bw, Ps, Pt, Wn2, bn2 = _f32(bw), _f32(Ps), _f32(Pt), _f32(Wn2), _f32(bn2)
    E, V = bw.shape
    msg = torch.empty(E, V, device=bw.device, dtype=torch.float32)
    if save:
        pre2 = torch.empty(E, V, device=bw.device, dtype=torch.float32)
        _call("lcao_edge_node_message_saving", _p(bw), _p(Ps), _p(Pt), _p(src_p), _p(dst_p), _p(Wn2), _p(bn2), E, V,
              _p(msg), _p(pre2), _stream())
        return msg, pre2
    _call("lcao_edge_node_message_" + precision if precision in LOWP else "lcao_edge_node_message", _p(bw), _p(Ps), _p(Pt),
          _p(src_p), _p(dst_p), _p(Wn2), _p(bn2), E, V, _p(msg), _stream())
    return msg


def node_column_hint(out_ptr, n_edges: int) -> Tensor:
    """(1 + n_nodes,) int32 on the device: word 0 says whether the per-node-column form of edge_node_message_agg pays for this
    edge list (1: consecutive nodes; 2: nodes grouped by degree, order in words 1..; 0: no)."""
    out_ptr = _i32(out_ptr)
    hint = torch.zeros(out_ptr.numel(), device=out_ptr.device, dtype=torch.int32)
    _call("lcao_node_column_hint", _p(out_ptr), out_ptr.numel() - 1, n_edges, _p(hint), _stream())
    return hint


def edge_node_message_agg(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, column_hint=None, precision="fp32", save=False):
    """agg (n_centres, V): edge_node_message + the per-centre segment sum in one kernel (msg is never materialised).
    out_ptr (n_centres + 1) int32; the first out_ptr[-1] rows of bw / src_p / dst_p are the centres' out-edges.
    column_hint: node_column_hint(out_ptr, E) of this edge list (device-side choice of the per-node-column form).
    precision in LOWP: the 16-bit MFMA forms; their fallback for uneven degrees goes through a message scratch."""
    bw, Ps, Pt, Wn2, bn2 = _f32(bw), _f32(Ps), _f32(Pt), _f32(Wn2), _f32(bn2)
    E, V = bw.shape
    n = out_ptr.numel() - 1
    agg = torch.empty(n, V, device=bw.device, dtype=torch.float32)
    if precision in LOWP:
        msg = torch.empty(E, V, device=bw.device, dtype=torch.float32)
        _call("lcao_edge_node_message_agg_" + precision, _p(bw), _p(Ps), _p(Pt), _p(_i32(src_p)), _p(_i32(dst_p)),
              _p(_i32(out_ptr)), _p(Wn2), _p(bn2), n, E, V, _p(column_hint) if column_hint is not None else None, _p(msg),
              _p(agg), _stream())
        return agg
    if save:      # (agg, pre2): pre2 = the pre-activation of the second f_node layer, kept for the reverse sweep (fp32 only)
        pre2 = torch.empty(E, V, device=bw.device, dtype=torch.float32)
        _call("lcao_edge_node_message_agg_saving", _p(bw), _p(Ps), _p(Pt), _p(_i32(src_p)), _p(_i32(dst_p)), _p(_i32(out_ptr)),
              _p(Wn2), _p(bn2), n, E, V, _p(column_hint) if column_hint is not None else None, _p(pre2), _p(agg), _stream())
        return agg, pre2
    _call("lcao_edge_node_message_agg", _p(bw), _p(Ps), _p(Pt), _p(_i32(src_p)), _p(_i32(dst_p)), _p(_i32(out_ptr)), _p(Wn2),
          _p(bn2), n, E, V, _p(column_hint) if column_hint is not None else None, _p(agg), _stream())
    return agg


def edge_two_body_agg(tbw, B, W3, Wb, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, column_hint=None, precision="fp32"):
    """agg (n_centres, V) of lcaonet.py:192-213: edge_lcao_weight followed by edge_node_message_agg.  precision "f16": one
    C-ABI call (lcao_edge_two_body_agg_f16) — with a node-column hint one fused launch that never materialises bw."""
    if precision != "f16":
        bw = edge_lcao_weight(tbw, B, W3, Wb, precision)
        return edge_node_message_agg(bw, Ps, Pt, src_p, dst_p, out_ptr, Wn2, bn2, column_hint, precision)
    tbw, B, W3, Wb, Ps, Pt, Wn2, bn2 = (_f32(t) for t in (tbw, B, W3, Wb, Ps, Pt, Wn2, bn2))
    E, V = B.shape
    n = out_ptr.numel() - 1
    dev = B.device
    agg = torch.empty(n, V, device=dev, dtype=torch.float32)
    bw = torch.empty(E, V, device=dev, dtype=torch.float32)      # scratch of the uneven-degree fallback only
    msg = torch.empty(E, V, device=dev, dtype=torch.float32)
    _call("lcao_edge_two_body_agg_f16", _p(tbw), _p(B), _p(W3), _p(Wb), _p(Ps), _p(Pt), _p(_i32(src_p)), _p(_i32(dst_p)),
          _p(_i32(out_ptr)), _p(Wn2), _p(bn2), n, E, V, _p(column_hint) if column_hint is not None else None, _p(bw), _p(msg),
          _p(agg), _stream())
    return agg


# --- activation-gradient kernels (autograd forces at inference; csrc/lcao_backward.hip) -------------------------
def message_bwd_fused(g_agg, src_p, dst_p, bw, pre2, Ps, Pt, Wn2T, want_g_pre2=False, want_g_pre1=True):
    """(g_bw, g_pre1|None, g_pre2|None): message product, second f_node layer and SiLU'(pre1) differentiated in one kernel.
    want_g_pre1=False: g_bw only (the first layer of a force sweep: its node tables do not depend on the positions)."""
    g_agg, bw, pre2, Ps, Pt, Wn2T = _f32(g_agg), _f32(bw), _f32(pre2), _f32(Ps), _f32(Pt), _f32(Wn2T)
    E, V = bw.shape
    g_bw, g_pre1 = torch.empty_like(bw), (torch.empty_like(bw) if want_g_pre1 else None)
    g_pre2 = torch.empty_like(bw) if want_g_pre2 else None
    _call("lcao_message_bwd_fused", _p(g_agg), _p(src_p), _p(dst_p), _p(bw), _p(pre2), _p(Ps), _p(Pt), _p(Wn2T), E, V,
          _p(g_bw), _p(g_pre1), _p(g_pre2), _stream())
    return g_bw, g_pre1, g_pre2


def lcao_weight_bwd_fused(g_bw, B, ft, WbT, W3T, want_g_ft=False):
    """(g_B, g_tbw, g_ft|None): basis_weight^T, the normalised combine and f_three^T differentiated in one kernel."""
    g_bw, B, ft, WbT, W3T = _f32(g_bw), _f32(B), _f32(ft), _f32(WbT), _f32(W3T)
    E, V = B.shape
    g_B, g_tbw = torch.empty_like(B), torch.empty_like(B)
    g_ft = torch.empty_like(B) if want_g_ft else None
    _call("lcao_lcao_weight_bwd_fused", _p(g_bw), _p(B), _p(ft), _p(WbT), _p(W3T), E, V, _p(g_B), _p(g_tbw), _p(g_ft),
          _stream())
    return g_B, g_tbw, g_ft


def act_bwd(g, pre, act):
    g, pre = _f32(g), _f32(pre)
    out = torch.empty_like(g)
    _call("lcao_act_bwd", _p(g), _p(pre), g.numel(), act, _p(out), _stream())
    return out


def pair_pre(Ps, Pt, src_p, dst_p):
    Ps, Pt = _f32(Ps), _f32(Pt)
    E, V = src_p.numel(), Ps.shape[1]
    pre1 = torch.empty(E, V, device=Ps.device, dtype=torch.float32)
    _call("lcao_pair_pre", _p(Ps), _p(Pt), _p(src_p), _p(dst_p), E, V, _p(pre1), _stream())
    return pre1


def message_bwd(g_agg, src_p, bw, pre2):
    g_agg, bw, pre2 = _f32(g_agg), _f32(bw), _f32(pre2)
    E, V = bw.shape
    g_bw, g_pre2 = torch.empty_like(bw), torch.empty_like(bw)
    _call("lcao_message_bwd", _p(g_agg), _p(src_p), _p(bw), _p(pre2), E, V, _p(g_bw), _p(g_pre2), _stream())
    return g_bw, g_pre2


def combine_bwd(g_lw, B, ft, Bv=None):
    """(g_B, g_ft), or (g_B, g_Bv, g_ft) with the valence term Bv and ft (E, 2V)."""
    g_lw, B, ft = _f32(g_lw), _f32(B), _f32(ft)
    E, V = B.shape
    if Bv is not None:
        Bv = _f32(Bv)
        g_B, g_Bv, g_ft = torch.empty_like(B), torch.empty_like(B), torch.empty_like(ft)
        _call("lcao_combine_bwd_valence", _p(g_lw), _p(B), _p(Bv), _p(ft), E, V, _p(g_B), _p(g_Bv), _p(g_ft), _stream())
        return g_B, g_Bv, g_ft
    g_B, g_ft = torch.empty_like(B), torch.empty_like(B)
    _call("lcao_combine_bwd", _p(g_lw), _p(B), _p(ft), E, V, _p(g_B), _p(g_ft), _stream())
    return g_B, g_ft


def triplet_bwd(g_tbw, Ag, gram, uvec_p, src_p, dst_p, in_ptr, in_pos, out_ptr, tbase, n_trip_dense, e_own, n_own,
                mfma: bool | None = None):
    g_tbw, Ag, gram, uvec_p = _f32(g_tbw), _f32(Ag), _f32(gram), _f32(uvec_p)
    E, n_l, V = Ag.shape
    dev = Ag.device
    all_centres = (n_own == out_ptr.numel() - 1) and e_own == E
    if mfma is None:
        mfma = V in FUSED_WIDTHS
    if mfma:
        # every edge is an in-edge of exactly one node: with all nodes as centres every output row is written; with
        # owned centres only (sharded graph) the rows of the other edges must read as zero
        new = torch.empty if all_centres else torch.zeros
        g_Ag, g_gram = new(E, n_l, V, device=dev), new(E, 10, device=dev)
        g_uq, g_up = new(E, 3, device=dev), new(E, 3, device=dev)
        _call("lcao_triplet_bwd_mfma", _p(g_tbw), _p(Ag), _p(gram), _p(uvec_p), _p(in_ptr), _p(in_pos), _p(out_ptr), n_own,
              E, V, n_l, _p(g_Ag), _p(g_gram), _p(g_uq), _p(g_up), _stream())
        return g_Ag, g_gram, g_uq, g_up
    g_Ag = torch.zeros_like(Ag)          # edges that are in no owned in-list keep a zero gradient
    g_gram = torch.zeros(E, 10, device=dev)
    g_uq = torch.zeros(E, 3, device=dev)
    g_up = torch.zeros(E, 3, device=dev)
    gc = torch.empty(max(n_trip_dense, 1), device=dev, dtype=torch.float32)
    n_in_total = E   # launch-size bound; the kernel reads the exact count in_ptr[n_own] itself
    _call("lcao_triplet_bwd", _p(g_tbw), _p(Ag), _p(gram), _p(uvec_p), _p(src_p), _p(dst_p), _p(in_ptr), _p(in_pos),
          _p(out_ptr), _p(tbase), E, e_own, n_own, n_in_total, V, n_l, _p(g_Ag), _p(g_gram), _p(g_uq), _p(g_up), _p(gc),
          _stream())
    return g_Ag, g_gram, g_uq, g_up


def pair_contract_bwd(Tf, rb, order, l_of_d, z, src_p, dst_p, gate, g_Ag, g_B, g_gram, e_own, n_l, zdim, g_rb,
                      want_gA: bool = False, want_gate: bool = True):
    Tf, rb, gate, g_Ag, g_B, g_gram = _f32(Tf), _f32(rb), _f32(gate), _f32(g_Ag), _f32(g_B), _f32(g_gram)
    z = _i64(z)
    E, D = rb.shape
    V = Tf.shape[-1]
    g_gate_part = torch.empty(E, V, device=rb.device, dtype=torch.float32) if want_gate else None
    g_A = torch.empty(E, n_l, V, device=rb.device, dtype=torch.float32) if want_gA else None
    _call("lcao_pair_contract_bwd", _p(Tf), _p(rb), _p(order), _p(l_of_d), _p(z), _p(src_p), _p(dst_p), _p(gate), _p(g_Ag),
          _p(g_B), _p(g_gram), E, e_own, D, V, n_l, zdim, _p(g_rb), _p(g_gate_part), _p(g_A), _stream())
    return (g_gate_part, g_A) if want_gA else g_gate_part


def wgrad(g, x):
    """dW = g^T x for a Dense layer applied to many rows (g (E, M), x (E, K)); the MFMA kernel needs M, K % 64 == 0,
    other shapes go to the vendor GEMM (torch.mm)."""
    g, x = _f32(g), _f32(x)
    E, M = g.shape
    K = x.shape[1]
    if M % 64 or K % 64:
        return torch.mm(g.t(), x)
    dW = torch.empty(M, K, device=g.device, dtype=torch.float32)
    nbytes = int(_lib.load().lcao_wgrad_workspace_bytes(E, M, K))
    ws = torch.empty(max(nbytes, 16), device=g.device, dtype=torch.uint8)
    _call("lcao_wgrad", _p(g), _p(x), E, M, K, _p(dW), _p(ws), nbytes, _stream())
    return dW


def pair_table_grad(g_A, rb, order, l_of_d, perm, chunk_ptr):
    g_A, rb = _f32(g_A), _f32(rb)
    E, n_l, V = g_A.shape
    D = rb.shape[1]
    n_chunks = chunk_ptr.numel() - 1
    partial = torch.empty(n_chunks, D, V, device=rb.device, dtype=torch.float32)
    _call("lcao_pair_table_grad", _p(g_A), _p(rb), _p(order), _p(l_of_d), _p(perm), _p(chunk_ptr), n_chunks, D, V, n_l,
          _p(partial), _stream())
    return partial


def geometry_bwd(dist, uvec, g_rb, g_u, orb_table, rbf_kind, cutoff_kind, rc):
    dist, uvec, g_rb, g_u, orb_table = _f32(dist), _f32(uvec), _f32(g_rb), _f32(g_u), _f32(orb_table)
    E, D = g_rb.shape
    g_vec = torch.empty(E, 3, device=dist.device, dtype=torch.float32)
    _call("lcao_geometry_bwd", _p(dist), _p(uvec), _p(g_rb), _p(g_u), E, D, _p(orb_table), RBF_KINDS[rbf_kind],
          CUTOFF_KINDS[cutoff_kind], float(rc), _p(g_vec), _stream())
    return g_vec


# --- three-body bilinear family with explicit per-triplet weights (csrc/lcao_double.hip; model/double.py) ----------
def trip_spmm(w, A, geo):
    """out[p,:] = sum_{q,l} w[t(p,q),l] A[q,l,:];  geo: model/double.py::_Geo (CSR + triplet slot bases)."""
    w, A = _f32(w, "w"), _f32(A, "A")
    _, n_l, V = A.shape
    c, E = geo.ctx, geo.e_own
    out = torch.empty(E, V, device=A.device, dtype=torch.float32)
    _call("lcao_trip_spmm", _p(w), _p(A), _p(c.src_p), _p(c.in_ptr), _p(c.in_pos), _p(c.out_ptr), _p(geo.tbase), E,
          geo.n_own, V, n_l, _p(out), _stream())
    return out


def trip_spmm_t(w, G, geo):
    """out[q,l,:] = sum_p w[t(p,q),l] G[p,:]"""
    w, G = _f32(w, "w"), _f32(G, "G")
    V = G.shape[1]
    n_l = w.shape[1]
    c, E = geo.ctx, geo.E
    out = torch.empty(E, n_l, V, device=G.device, dtype=torch.float32)
    _call("lcao_trip_spmm_t", _p(w), _p(G), _p(c.dst_p), _p(geo.in_rank), _p(c.in_ptr), _p(c.in_pos), _p(c.out_ptr),
          _p(geo.tbase), E, geo.n_own, V, n_l, int(geo.n_own == geo.N), _p(out), _stream())
    return out


def trip_sddmm(G, A, geo):
    """out[t,l] = G[p,:] . A[q,l,:]"""
    G, A = _f32(G, "G"), _f32(A, "A")
    _, n_l, V = A.shape
    c, E = geo.ctx, geo.e_own
    out = torch.zeros(geo.T, n_l, device=A.device, dtype=torch.float32)
    _call("lcao_trip_sddmm", _p(G), _p(A), _p(c.src_p), _p(c.in_ptr), _p(c.in_pos), _p(c.out_ptr), _p(geo.tbase), E,
          geo.n_own, V, n_l, _p(out), _stream())
    return out


def _tw_args(geo):
    c = geo.ctx
    return (_p(c.dst_p), _p(c.in_ptr), _p(c.in_pos), _p(c.out_ptr), _p(geo.tbase), geo.E, geo.n_own)


def _tw_check(c, G, geo):
    n_l = G.shape[1]
    if tuple(c.shape) != (geo.T,) or tuple(G.shape) != (geo.E, n_l, n_l):
        raise ValueError(f"trip_weight: c {tuple(c.shape)} must be ({geo.T},) and G {tuple(G.shape)} (E={geo.E}, n_l, n_l)")
    return n_l


def trip_weight(c, G, geo):
    """w[t,l] = [p != q] Y_l(c_t) / sqrt(max(Y^T G_q Y, 1e-24));  c (T,), G (E, n_l, n_l) -> (T, n_l)."""
    c, G = _f32(c, "c"), _f32(G, "G")
    n_l = _tw_check(c, G, geo)
    w = torch.zeros(geo.T, n_l, device=c.device, dtype=torch.float32)
    _call("lcao_trip_weight", _p(c), _p(G), *_tw_args(geo), n_l, _p(w), _stream())
    return w


def trip_weight_bwd(gw, c, G, geo):
    """(gc (T,), gG (E, n_l, n_l)): the reverse sweep of trip_weight."""
    gw, c, G = _f32(gw, "gw"), _f32(c, "c"), _f32(G, "G")
    n_l = _tw_check(c, G, geo)
    gc = torch.zeros(geo.T, device=c.device, dtype=torch.float32)
    gG = torch.empty_like(G)
    _call("lcao_trip_weight_bwd", _p(gw), _p(c), _p(G), *_tw_args(geo), n_l, _p(gc), _p(gG), _stream())
    return gc, gG


def trip_weight_bwd2(hc, hG, gw, c, G, geo):
    """Reverse sweep of trip_weight_bwd for cotangents (hc, hG) of (gc, gG): (ggw (T, n_l), gc2 (T,), gG2 (E, n_l, n_l))."""
    hc, hG, gw, c, G = _f32(hc, "hc"), _f32(hG, "hG"), _f32(gw, "gw"), _f32(c, "c"), _f32(G, "G")
    n_l = _tw_check(c, G, geo)
    ggw = torch.zeros(geo.T, n_l, device=c.device, dtype=torch.float32)
    gc2 = torch.zeros(geo.T, device=c.device, dtype=torch.float32)
    gG2 = torch.empty_like(G)
    _call("lcao_trip_weight_bwd2", _p(hc), _p(hG), _p(gw), _p(c), _p(G), *_tw_args(geo), n_l, _p(ggw), _p(gc2), _p(gG2),
          _stream())
    return ggw, gc2, gG2


def radial_levels(level: int, r, orb_table, rbf_kind: str, cutoff_kind: str, rc: float, g=None, h=None):
    """Radial basis with R', R'' (csrc/lcao_double.hip): level 0 -> rb (E, D); 1 -> gr (E,) = sum_d g R';
    2 -> (gg (E, D) = h R', gr2 (E,) = h sum_d g R'')."""
    r, orb_table = _f32(r, "r"), _f32(orb_table, "orb_table")
    E, D = r.numel(), orb_table.shape[0]
    g = None if g is None else _f32(g, "g")
    h = None if h is None else _f32(h, "h")
    if g is not None and tuple(g.shape) != (E, D):
        raise ValueError(f"radial_levels: g {tuple(g.shape)} must be ({E}, {D})")
    if h is not None and h.numel() != E:
        raise ValueError(f"radial_levels: h {tuple(h.shape)} must have {E} elements")
    o_ed = torch.empty(E, D, device=r.device, dtype=torch.float32) if level != 1 else None
    o_e = torch.empty(E, device=r.device, dtype=torch.float32) if level >= 1 else None
    opt = lambda t: _p(t) if t is not None else None
    _call("lcao_radial_levels", level, _p(r), opt(g), opt(h), E, D, _p(orb_table), RBF_KINDS[rbf_kind],
          CUTOFF_KINDS[cutoff_kind], float(rc), opt(o_ed), opt(o_e), _stream())
    return o_ed if level == 0 else (o_e if level == 1 else (o_ed, o_e))


def silu_levels(level: int, x, g=None, h=None):
    """SiLU with closed-form derivatives (csrc/lcao_double.hip): level 0 -> y; 1 -> g s1(x); 2 -> (h s1(x), h g s2(x))."""
    x = _f32(x, "x")
    g = None if g is None else _f32(g, "g")
    h = None if h is None else _f32(h, "h")
    for t in (g, h):
        if t is not None and t.shape != x.shape:
            raise ValueError("silu_levels: operands must have one shape")
    o0 = torch.empty_like(x)
    o1 = torch.empty_like(x) if level == 2 else None
    _call("lcao_silu_levels", level, _p(x), _p(g) if g is not None else None, _p(h) if h is not None else None,
          x.numel(), _p(o0), _p(o1) if o1 is not None else None, _stream())
    return o0 if level < 2 else (o0, o1)


def bcast_fma(X1, y1, X2=None, y2=None, z=None):
    """out[e,l,:] = X1[e,l,:] y1[e,:] (+ X2[e,l,:] y2[e,:]) (+ z[e,:]);  X (E, L, V), y / z (E, V)."""
    X1, y1 = _f32(X1, "X1"), _f32(y1, "y1")
    E, L, V = X1.shape
    opt = lambda t, shape: None if t is None else _f32(t)
    X2, y2, z = opt(X2, None), opt(y2, None), opt(z, None)
    for t, shape in ((y1, (E, V)), (X2, (E, L, V)), (y2, (E, V)), (z, (E, V))):
        if t is not None and tuple(t.shape) != shape:
            raise ValueError(f"bcast_fma: operand {tuple(t.shape)} where {shape} is expected")
    out = torch.empty_like(X1)
    p = lambda t: _p(t) if t is not None else None
    _call("lcao_bcast_fma", _p(X1), _p(y1), p(X2), p(y2), p(z), E, L, V, _p(out), _stream())
    return out


def lsum_mul(X, Y=None):
    """out[e,:] = sum_l X[e,l,:] (Y[e,l,:] or 1)."""
    X = _f32(X, "X")
    E, L, V = X.shape
    Y = None if Y is None else _f32(Y, "Y")
    if Y is not None and Y.shape != X.shape:
        raise ValueError("lsum_mul: X and Y must have one shape")
    out = torch.empty(E, V, device=X.device, dtype=torch.float32)
    _call("lcao_lsum_mul", _p(X), _p(Y) if Y is not None else None, E, L, V, _p(out), _stream())
    return out


def mulp_levels(level: int, a, b, c0: float, g=None, ha=None, hb=None):
    """a (c0 + b) with closed-form derivative levels (csrc/lcao_double.hip): 0 -> f; 1 -> (g (c0 + b), g a);
    2 -> (ha (c0 + b) + hb a, hb g, ha g)."""
    a, b = _f32(a, "a"), _f32(b, "b")
    ts = [None if t is None else _f32(t) for t in (g, ha, hb)]
    for t in [b] + [t for t in ts if t is not None]:
        if t.shape != a.shape:
            raise ValueError("mulp_levels: operands must have one shape")
    outs = [torch.empty_like(a) for _ in range(level + 1)]
    opt = lambda t: _p(t) if t is not None else None
    po = [_p(o) for o in outs] + [None] * (2 - level)
    _call("lcao_mulp_levels", level, _p(a), _p(b), opt(ts[0]), opt(ts[1]), opt(ts[2]), float(c0), a.numel(), po[0], po[1], po[2],
          _stream())
    return outs[0] if level == 0 else tuple(outs)


def _three_level_call(name, level, a, b, cots, extra):
    a, b = _f32(a, "a"), _f32(b, "b")
    ts = [None if t is None else _f32(t) for t in cots]
    for t in [b] + [t for t in ts if t is not None]:
        if t.shape != a.shape:
            raise ValueError(f"{name}: operands must have one shape")
    outs = [torch.empty_like(a) for _ in range(level + 1)]
    opt = lambda t: _p(t) if t is not None else None
    po = [_p(o) for o in outs] + [None] * (2 - level)
    _call(name, level, _p(a), _p(b), opt(ts[0]), opt(ts[1]), opt(ts[2]), *extra(a), po[0], po[1], po[2], _stream())
    return outs[0] if level == 0 else tuple(outs)


def injnorm_levels(level: int, a, b, g=None, ha=None, hb=None, eps: float = 1e-12):
    """F.normalize(a (1 + b), dim=-1) with derivative levels (csrc/lcao_double.hip::k_injnorm_levels): 0 -> y;
    1 -> (d/da, d/db); 2 -> (d/dg, d/da, d/db)."""
    V = a.shape[-1]
    return _three_level_call("lcao_injnorm_levels", level, a, b, (g, ha, hb), lambda t: (t.numel() // max(V, 1), V, float(eps)))


def silumul_levels(level: int, a, x, g=None, ha=None, hx=None):
    """a * silu(x) with derivative levels (csrc/lcao_double.hip::k_silumul_levels): 0 -> m; 1 -> (d/da, d/dx);
    2 -> (d/dg, d/da, d/dx).  numel % 4 == 0."""
    return _three_level_call("lcao_silumul_levels", level, a, x, (g, ha, hx), lambda t: (t.numel(),))


def pairsilu_levels(level: int, Ps, Pt, src_p, dst_p, g=None, hPs=None, hPt=None):
    """silu(Ps[src] + Pt[dst]) with derivative levels (csrc/lcao_double.hip::k_pairsilu_levels): 0 -> h1; 1 -> dE/dpre;
    2 -> (d/dg, d/dpre).  The segment sums to the node tables are the caller's."""
    Ps, Pt = _f32(Ps, "Ps"), _f32(Pt, "Pt")
    E, V = src_p.numel(), Ps.shape[1]
    opt = lambda t: _p(_f32(t)) if t is not None else None
    o0 = torch.empty(E, V, device=Ps.device, dtype=torch.float32)
    o1 = torch.empty_like(o0) if level == 2 else None
    _call("lcao_pairsilu_levels", level, _p(Ps), _p(Pt), _p(src_p), _p(dst_p), opt(g), opt(hPs), opt(hPt), E, V, _p(o0),
          _p(o1) if o1 is not None else None, _stream())
    return o0 if level < 2 else (o0, o1)


ASIDE_WIDTHS = (32, 64, 128)


def aside_levels(level: int, Tf, rb, l_of_d, z, src_p, dst_p, gate, n_l: int, zdim: int, gAg=None, gB=None, gG=None,
                 h_rb=None, h_gate=None, want_table_terms: bool = False):
    """The orbital contraction with its consumers and its derivative levels (csrc/lcao_double.hip::k_aside):
    0 -> (Ag, B, G);  1 -> (g_rb, g_gate_part[, gA]);  2 -> (d_gAg, d_gB, d_gG, d_rb, d_gate_part[, gA, dA])."""
    Tf, rb, gate = _f32(Tf, "Tf"), _f32(rb, "rb"), _f32(gate, "gate")
    E, D = rb.shape
    V = Tf.shape[-1]
    if Tf.shape[0] != zdim * zdim or Tf.shape[1] != D or gate.shape[1] != V:
        raise ValueError("aside_levels: Tf (zdim^2, D, V), rb (E, D), gate (N, V)")
    dev = rb.device
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    opt = lambda t: _p(t) if t is not None else None
    if level >= 1:
        gAg, gB, gG = _f32(gAg, "gAg"), _f32(gB, "gB"), _f32(gG, "gG")
        if gAg.shape != (E, n_l, V) or gB.shape != (E, V) or gG.shape != (E, n_l, n_l):
            raise ValueError("aside_levels: cotangents gAg (E, n_l, V), gB (E, V), gG (E, n_l, n_l)")
    if level == 2:
        h_rb = _f32(h_rb, "h_rb")
        h_gate = None if h_gate is None else _f32(h_gate, "h_gate")
        if h_rb.shape != (E, D) or (h_gate is not None and h_gate.shape != gate.shape):
            raise ValueError("aside_levels: cotangents h_rb (E, D), h_gate (N, V)")
    big0 = new(E, n_l, V) if level != 1 else None
    row0 = new(E, V) if level != 1 else None
    gram = new(E, n_l, n_l) if level != 1 else None
    o_rb = new(E, D) if level >= 1 else None
    row1 = new(E, V) if level >= 1 else None
    o_gA = new(E, n_l, V) if (level >= 1 and want_table_terms) else None
    o_dA = new(E, n_l, V) if (level == 2 and want_table_terms) else None
    _call("lcao_aside_levels", level, _p(Tf), _p(rb), _p(l_of_d), _p(z), _p(src_p), _p(dst_p), _p(gate), opt(gAg), opt(gB),
          opt(gG), opt(h_rb), opt(h_gate), E, D, V, n_l, zdim, opt(big0), opt(row0), opt(gram), opt(o_rb), opt(row1),
          opt(o_gA), opt(o_dA), _stream())
    if level == 0:
        return big0, row0, gram
    if level == 1:
        return (o_rb, row1, o_gA) if want_table_terms else (o_rb, row1)
    return (big0, row0, gram, o_rb, row1, o_gA, o_dA) if want_table_terms else (big0, row0, gram, o_rb, row1)


ASIDE_PAIRS_MAX_ORB = 24


def aside_pairs(level: int, Tf, rb, l_of_d, z, src_p, dst_p, gate, n_l: int, zdim: int, perm, chunk_ptr, gAg, gB, gG,
                h_rb=None, h_gate=None, order=None, gram_tri: bool = False):
    """Levels 1 / 2 of aside_levels with the per-chunk parts of the table gradient (csrc/lcao_double.hip::k_aside_pairs):
    1 -> (g_rb, g_gate_part, partial);  2 -> (d_gAg, d_gB, d_gG, d_rb, d_gate_part, partial);  partial (n_chunks, D, V)."""
    Tf, rb, gate = _f32(Tf, "Tf"), _f32(rb, "rb"), _f32(gate, "gate")
    gAg, gB, gG = _f32(gAg, "gAg"), _f32(gB, "gB"), _f32(gG, "gG")
    E, D = rb.shape
    V = Tf.shape[-1]
    if (gAg.shape != (E, n_l, V) or gB.shape != (E, V) or gG.shape != ((E, 10) if gram_tri else (E, n_l, n_l))
            or Tf.shape[:2] != (zdim * zdim, D)):
        raise ValueError("aside_pairs: gAg (E, n_l, V), gB (E, V), gG (E, n_l, n_l) or (E, 10), Tf (zdim^2, D, V)")
    dev = rb.device
    new = lambda *shape: torch.empty(*shape, device=dev, dtype=torch.float32)
    opt = lambda t: _p(t) if t is not None else None
    if level == 2:
        h_rb = _f32(h_rb, "h_rb")
        h_gate = None if h_gate is None else _f32(h_gate, "h_gate")
    n_chunks = chunk_ptr.numel() - 1
    big0 = new(E, n_l, V) if level == 2 else None
    row0 = new(E, V) if level == 2 else None
    gram = new(E, n_l, n_l) if level == 2 else None
    o_rb, row1, partial = new(E, D), new(E, V), new(n_chunks, D, V)
    _call("lcao_aside_pairs", level, _p(Tf), _p(rb), _p(l_of_d), _p(z), _p(src_p), _p(dst_p), _p(gate), _p(gAg), _p(gB),
          _p(gG), opt(h_rb), opt(h_gate), _p(perm), _p(chunk_ptr), opt(order), int(gram_tri), n_chunks, E, D, V, n_l, zdim,
          opt(big0), opt(row0),
          opt(gram), _p(o_rb), _p(row1), _p(partial), _stream())
    return (o_rb, row1, partial) if level == 1 else (big0, row0, gram, o_rb, row1, partial)


def rownorm_levels(level: int, x, g=None, k=None, eps: float = 1e-12):
    """x / max(|x|, eps) along the last dim with closed-form derivatives; levels as silu_levels."""
    x = _f32(x, "x")
    g = None if g is None else _f32(g, "g")
    k = None if k is None else _f32(k, "k")
    for t in (g, k):
        if t is not None and t.shape != x.shape:
            raise ValueError("rownorm_levels: operands must have one shape")
    V = x.shape[-1]
    o0 = torch.empty_like(x)
    o1 = torch.empty_like(x) if level == 2 else None
    _call("lcao_rownorm_levels", level, _p(x), _p(g) if g is not None else None, _p(k) if k is not None else None,
          x.numel() // max(V, 1), V, float(eps), _p(o0), _p(o1) if o1 is not None else None, _stream())
    return o0 if level < 2 else (o0, o1)


def rowgram(X, Y):
    """out[e,l,m] = sum_v X[e,l,v] Y[e,m,v]  — per-edge Gram matrices (E, n_l, n_l)."""
    X, Y = _f32(X, "X"), _f32(Y, "Y")
    if X.shape != Y.shape or X.dim() != 3:
        raise ValueError(f"rowgram: X {tuple(X.shape)} and Y {tuple(Y.shape)} must both be (E, n_l, V)")
    E, n_l, V = X.shape
    out = torch.empty(E, n_l, n_l, device=X.device, dtype=torch.float32)
    _call("lcao_rowgram", _p(X), _p(Y), E, V, n_l, _p(out), _stream())
    return out


def rowmix(S, A, trans: bool = False):
    """out[e,l,v] = sum_m S[e,l,m] A[e,m,v]   (S[e,m,l] with trans)."""
    S, A = _f32(S, "S"), _f32(A, "A")
    E, n_l, V = A.shape
    if tuple(S.shape) != (E, n_l, n_l):
        raise ValueError(f"rowmix: S {tuple(S.shape)} must be ({E}, {n_l}, {n_l})")
    out = torch.empty(E, n_l, V, device=A.device, dtype=torch.float32)
    _call("lcao_rowmix", _p(S), _p(A), E, V, n_l, int(bool(trans)), _p(out), _stream())
    return out
