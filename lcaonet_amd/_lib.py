"""ctypes binding of liblcaonet_hip.so (the C ABI declared in include/lcaonet_hip.h).

The product path has NO fallback: if the library is missing or fails to load, every op raises."""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "liblcaonet_hip.so")

P, I, L, F = c_void_p, c_int, c_int64, c_float

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/lcaonet_hip.h one to one
SIGNATURES = {
    "lcao_abi_version": [],
    "lcao_error_string": [I],
    "lcao_edge_geometry": [P, P, P, P, P, P, L, P, P, P],
    "lcao_cutoff": [P, L, I, F, P, P],
    "lcao_radial_basis": [P, L, I, P, I, I, F, P, P],
    "lcao_edge_geometry_basis": [P, P, P, P, P, P, P, L, I, P, I, I, F, P, P, P, P, P],
    "lcao_sph_basis": [P, L, I, P, P, P],
    "lcao_triplet_angles": [P, P, P, L, P, P],
    "lcao_build_csr_workspace_bytes": [L, L],
    "lcao_build_csr": [P, P, L, L, P, P, P, P, P, P, P, P, P, c_size_t, P, P],
    "lcao_checked_index": [P, L, L, L, I, P, P, P],
    "lcao_nl_workspace_bytes": [L],
    "lcao_nl_bin_sort": [P, P, P, L, L, P, P, P, P, P, c_size_t, P],
    "lcao_nl_search": [P, P, P, P, P, P, L, F, I, P, P, P, P, P, P, P],
    "lcao_dense": [P, L, P, P, P, L, P, L, L, I, I, I, P],
    "lcao_pair_rows": [P, P, P, L, I, P, P],
    "lcao_node_tables": [P, P, P, L, I, I, P, P, P, P],
    "lcao_segment_sum": [P, P, P, L, I, P, P],
    "lcao_segment_sum_gather": [P, P, P, L, I, P, P],
    "lcao_gather_rows": [P, P, L, I, P, P],
    "lcao_permute_rows": [P, P, L, I, P, P],
    "lcao_coeff_contract": [P, P, P, P, P, L, L, L, L, I, I, I, I, P, P, P, P],
    "lcao_triplet_sum": [P, P, P, P, P, P, P, L, I, I, P, P],
    "lcao_edge_combine": [P, P, P, L, I, P, P],
    "lcao_pair_silu": [P, P, P, P, L, I, P, P],
    "lcao_embed_coeffs": [P, P, P, P, P, P, P, P, L, I, I, P, P],
    "lcao_triplet_literal": [P, P, P, P, L, P, P, P, L, L, L, L, I, I, I, P, P],
    "lcao_coeff_mlp_contract": [P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, I, P, P, P, P, P, P, c_size_t, P],
    "lcao_coeff_mlp_contract_bf16x3": [P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, P, P, P, c_size_t, P],
    "lcao_coeff_mlp_contract_f16": [P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, P, P, P, c_size_t, P],
    "lcao_lowp_packed_bytes": [L, I],
    "lcao_lowp_pack_rows": [P, L, I, P, P],
    "lcao_edge_lcao_weight_f16": [P, P, P, P, L, I, P, P],
    "lcao_edge_node_message_f16": [P, P, P, P, P, P, P, L, I, P, P],
    "lcao_triplet_gemm": [P, P, P, P, P, P, L, L, I, I, P, P],
    "lcao_pair_contract": [P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, P, P],
    "lcao_pair_contract_visit": [P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, P, P],
    "lcao_edge_lcao_weight": [P, P, P, P, L, I, P, P],
    "lcao_edge_node_message": [P, P, P, P, P, P, P, L, I, P, P],
    "lcao_edge_node_message_agg": [P, P, P, P, P, P, P, P, L, L, I, P, P, P],
    "lcao_edge_node_message_agg_saving": [P, P, P, P, P, P, P, P, L, L, I, P, P, P, P],
    "lcao_node_column_hint": [P, L, L, P, P],
    "lcao_edge_node_message_agg_bf16x3": [P, P, P, P, P, P, P, P, L, L, I, P, P, P, P],
    "lcao_edge_node_message_agg_f16": [P, P, P, P, P, P, P, P, L, L, I, P, P, P, P],
    "lcao_edge_two_body_agg_f16": [P, P, P, P, P, P, P, P, P, P, P, L, L, I, P, P, P, P, P],
    "lcao_edge_lcao_weight_saving": [P, P, P, P, L, I, P, P, P],
    "lcao_edge_node_message_saving": [P, P, P, P, P, P, P, L, I, P, P, P],
    "lcao_wgrad": [P, P, L, I, I, P, P, c_size_t, P],
    "lcao_wgrad_workspace_bytes": [L, I, I],
    "lcao_combine_bwd_valence": [P, P, P, P, L, I, P, P, P, P],
    "lcao_message_bwd_fused": [P, P, P, P, P, P, P, P, L, I, P, P, P, P],
    "lcao_lcao_weight_bwd_fused": [P, P, P, P, P, L, I, P, P, P, P],
    "lcao_edge_lcao_weight_bf16x3": [P, P, P, P, L, I, P, P],
    "lcao_edge_node_message_bf16x3": [P, P, P, P, P, P, P, L, I, P, P],
    "lcao_act_bwd": [P, P, L, I, P, P],
    "lcao_pair_pre": [P, P, P, P, L, I, P, P],
    "lcao_message_bwd": [P, P, P, P, L, I, P, P, P],
    "lcao_combine_bwd": [P, P, P, L, I, P, P, P],
    "lcao_triplet_bwd": [P, P, P, P, P, P, P, P, P, P, L, L, I, L, I, I, P, P, P, P, P, P],
    "lcao_triplet_bwd_mfma": [P, P, P, P, P, P, P, L, L, I, I, P, P, P, P, P],
    "lcao_pair_contract_bwd": [P, P, P, P, P, P, P, P, P, P, P, L, L, I, I, I, I, P, P, P, P],
    "lcao_pair_table_grad": [P, P, P, P, P, P, L, I, I, I, P, P],
    "lcao_geometry_bwd": [P, P, P, P, L, I, P, I, I, F, P, P],
    "lcao_trip_spmm": [P, P, P, P, P, P, P, L, L, I, I, P, P],
    "lcao_trip_spmm_t": [P, P, P, P, P, P, P, P, L, L, I, I, I, P, P],
    "lcao_trip_sddmm": [P, P, P, P, P, P, P, L, L, I, I, P, P],
    "lcao_trip_weight": [P, P, P, P, P, P, P, L, L, I, P, P],
    "lcao_trip_weight_bwd": [P, P, P, P, P, P, P, P, L, L, I, P, P, P],
    "lcao_trip_weight_bwd2": [P, P, P, P, P, P, P, P, P, P, L, L, I, P, P, P, P],
    "lcao_radial_levels": [I, P, P, P, L, I, P, I, I, F, P, P, P],
    "lcao_mulp_levels": [I, P, P, P, P, P, F, L, P, P, P, P],
    "lcao_aside_levels": [I, P, P, P, P, P, P, P, P, P, P, P, P, L, I, I, I, I, P, P, P, P, P, P, P, P],
    "lcao_injnorm_levels": [I, P, P, P, P, P, L, I, F, P, P, P, P],
    "lcao_silumul_levels": [I, P, P, P, P, P, L, P, P, P, P],
    "lcao_pairsilu_levels": [I, P, P, P, P, P, P, P, L, I, P, P, P],
    "lcao_aside_pairs": [I, P, P, P, P, P, P, P, P, P, P, P, P, P, P, P, I, L, L, I, I, I, I, P, P, P, P, P, P, P],
    "lcao_bcast_fma": [P, P, P, P, P, L, I, I, P, P],
    "lcao_lsum_mul": [P, P, L, I, I, P, P],
    "lcao_silu_levels": [I, P, P, P, L, P, P, P],
    "lcao_rownorm_levels": [I, P, P, P, L, I, F, P, P, P],
    "lcao_rowgram": [P, P, L, I, I, P, P],
    "lcao_rowmix": [P, P, L, I, I, I, P, P],
    "lcao_lds_bytes_coeff_mlp": [I, I, I],
    "lcao_debug_set_clock_probe": [P],
}
ABI_VERSION = 2
_RESTYPES = {"lcao_error_string": c_char_p, "lcao_lds_bytes_coeff_mlp": c_size_t, "lcao_lowp_packed_bytes": c_size_t,
             "lcao_build_csr_workspace_bytes": c_size_t, "lcao_nl_workspace_bytes": c_size_t, "lcao_wgrad_workspace_bytes": c_size_t, "lcao_debug_set_clock_probe": None}

_lib = None


class LcaoHipError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the library once; raise loudly if it is absent (no CPU fallback exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LcaoHipError(
            f"{LIB_PATH} not found: build it with `python -m lcaonet_amd.build` (hipcc, gfx950). "
            "lcaonet_amd has no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the header and the library drift apart
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.lcao_abi_version() != ABI_VERSION:
        raise LcaoHipError(f"ABI version mismatch: library {lib.lcao_abi_version()}, binding {ABI_VERSION}")
    _lib = lib
    return lib


def check(code: int, what: str) -> None:
    if code != 0:
        msg = load().lcao_error_string(code).decode()
        raise LcaoHipError(f"{what} failed: {msg} (code {code})")
