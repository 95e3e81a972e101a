"""CPU-side checks (no GPU): tables, resolvers, state_dict layout, C-ABI surface, loud failure off-GPU."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

from conftest import MODEL_CASES, REPO, Golden

from lcaonet_amd import _lib
from lcaonet_amd.atomistic import elec
from lcaonet_amd.atomistic.info import ElecInfo
from lcaonet_amd.model import LCAONet
from lcaonet_amd.nn.rbf import laguerre_coeffs, orbital_table
from lcaonet_amd.utils.resolve import activation_resolver, cutoffnet_resolver, init_resolver, rbf_resolver


def test_tables_match_reference(golden_loader):
    z = golden_loader("atomistic_tables").z
    assert np.array_equal(elec.ELEC_TABLE.numpy(), z["ELEC_TABLE"])
    assert np.array_equal(elec.VALENCE_TABLE.numpy(), z["VALENCE_TABLE"])
    assert np.array_equal(elec.MAX_ELEC_IDX.numpy(), z["MAX_ELEC_IDX"])
    assert np.array_equal(elec.NL_LIST.numpy(), z["NL_LIST"])


def test_elecinfo_matches_reference(golden_loader):
    z = golden_loader("atomistic_tables").z
    for c in json.loads(str(z["elecinfo_cases"])):
        ei = ElecInfo(c["max_z"], c["max_orb"], c["min_orb"], c["n_per_orb"])
        assert ei.n_orb == c["n_orb"] and ei.min_orb_idx == c["min_orb_idx"]
        assert ei.nl_list.tolist() == c["nl"] and ei.max_elec_idx.tolist() == c["max_elec_idx"]
        assert int(ei.elec_table.sum()) == c["elec_sum"] and int(ei.valence_table.sum()) == c["val_sum"]


def test_elecinfo_errors():
    for bad in (0, -3, 97):
        with pytest.raises(ValueError):
            ElecInfo(bad, None)
    with pytest.raises(ValueError):
        ElecInfo(10, "9z")


def test_laguerre_tables():
    # SURVEY.md §8a5 closed forms (probe of the reference's lambdified polynomials), ascending powers here
    assert laguerre_coeffs(1, 0) == [-1]
    assert laguerre_coeffs(2, 0) == [-4, 2]
    assert laguerre_coeffs(3, 0) == [-18, 18, -3]
    assert laguerre_coeffs(4, 1) == [-1200, 600, -60]
    assert laguerre_coeffs(7, 0) == [-35280, 105840, -88200, 29400, -4410, 294, -7]
    assert laguerre_coeffs(6, 2) == [-2257920, 1128960, -161280, 6720]
    t = orbital_table([(1, 0), (7, 0), (4, 3)])
    assert t.shape == (3, 12)
    assert abs(float(t[0, 1]) - (-5.198122233)) < 1e-6 and abs(float(t[2, 1]) - (-9.079892364e-7)) < 1e-12


def test_resolvers():
    assert type(activation_resolver("SiLU")) is torch.nn.SiLU
    assert type(activation_resolver("shifted-softplus")).__name__ == "ShiftedSoftplus"
    assert init_resolver("glorotorthogonal").__name__ == "glorot_orthogonal"
    assert type(cutoffnet_resolver("envelope", cutoff=5.0)).__name__ == "EnvelopeCutoff"
    assert type(cutoffnet_resolver("Polynomial_Cutoff", cutoff=5.0)).__name__ == "PolynomialCutoff"
    ei = ElecInfo(10, None)
    assert type(rbf_resolver("hydrogen", cutoff=5.0, elec_info=ei, cutoff_net=cutoffnet_resolver("cosine", cutoff=5.0))
                ).__name__ == "HydrogenRadialBasis"
    with pytest.raises(ValueError):
        cutoffnet_resolver("nosuch", cutoff=1.0)
    with pytest.raises(ValueError):
        activation_resolver("nosuchact")


def test_activation_gain_resolver():
    """reference utils/resolve.py:79-105"""
    from lcaonet_amd.utils.resolve import activation_gain_resolver as gain
    assert gain("relu") == torch.nn.init.calculate_gain("relu")
    assert gain("Leaky_ReLU", param=0.2) == torch.nn.init.calculate_gain("leaky_relu", 0.2)
    assert gain("swish") == gain(torch.nn.Sigmoid) == 1.0
    assert gain("shifted-softplus") == gain("silu") == gain("no such activation") == 1.0
    assert gain("tanh") == pytest.approx(5.0 / 3.0)


@pytest.mark.parametrize("name", MODEL_CASES)
def test_state_dict_is_drop_in(golden_loader, name):
    """A reference checkpoint loads with strict=True: identical parameter / buffer names and shapes."""
    g = golden_loader(name)
    kw = {k: v for k, v in g.cfg.items() if k != "training"}
    for k in ("atomref", "mean"):
        if kw.get(k) is not None:
            kw[k] = torch.tensor(kw[k])
    m = LCAONet(**kw)
    res = m.load_state_dict(g.state_dict(), strict=True)
    assert not res.missing_keys and not res.unexpected_keys


def test_readme_kwarg_spellings():
    m = LCAONet(hidden_dim=16, coeffs_dim=32, conv_dim=16, out_dim=2, n_interaction=1)
    assert (m.emb_size, m.emb_size_coeff, m.emb_size_conv, m.out_size) == (16, 32, 16, 2)


def test_c_abi_exports_every_declared_symbol():
    """The shared library loads without a GPU and exports exactly what include/lcaonet_hip.h declares."""
    hdr = open(os.path.join(REPO, "include", "lcaonet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(lcao_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 20
    assert os.path.exists(_lib.LIB_PATH), "build with python -m lcaonet_amd.build"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for sym in declared:
        assert hasattr(lib, sym), f"{sym} declared in the header but not exported"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert _lib.load().lcao_abi_version() == _lib.ABI_VERSION == 2
    assert b"invalid" in _lib.load().lcao_error_string(-1)


def test_ctypes_signatures_match_the_header_prototypes():
    """Every prototype in include/lcaonet_hip.h against lcaonet_amd/_lib.py::SIGNATURES: same number of parameters and the
    same C type class per position (pointer / int64_t / int / float / size_t) — an ABI mismatch would otherwise only show
    up as a corrupted argument on the GPU."""
    from ctypes import c_float, c_int, c_int64, c_size_t, c_void_p
    hdr = open(os.path.join(REPO, "include", "lcaonet_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    protos = re.findall(r"\b(lcao_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.S)
    assert len(protos) >= 40

    def kind(param: str):
        param = " ".join(param.split())
        if "*" in param:
            return c_void_p
        if param.startswith("int64_t"):
            return c_int64
        if param.startswith("size_t"):
            return c_size_t
        if param.startswith("float"):
            return c_float
        if param.startswith("int"):
            return c_int
        raise AssertionError(f"unrecognised parameter {param!r}")

    checked = 0
    for name, params in protos:
        params = params.strip()
        want = [] if params in ("", "void") else [kind(p) for p in params.split(",")]
        got = list(_lib.SIGNATURES[name])
        assert len(got) == len(want), (name, len(got), len(want))
        for i, (g, w) in enumerate(zip(got, want)):
            assert g is w, (name, i, g, w)
        checked += 1
    assert checked == len(_lib.SIGNATURES)


def test_no_cpu_fallback():
    """Off-GPU the product path raises instead of computing on the host."""
    from lcaonet_amd import ops
    from lcaonet_amd.data.synthetic import reference_fixture_graph

    with pytest.raises(_lib.LcaoHipError):
        ops.dense(torch.zeros(4, 16), torch.zeros(16, 16))
    m = LCAONet(emb_size=16, emb_size_coeff=16, emb_size_conv=16, n_interaction=1).eval()
    with pytest.raises(_lib.LcaoHipError):
        m(reference_fixture_graph())


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "lcaonet_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", "") or "import oracle" not in src


def test_synthetic_generators():
    from lcaonet_amd.data.synthetic import jittered_cubic_cells, random_molecules, random_periodic_cells

    g = jittered_cubic_cells(3, 4, 3.2, 0.05, 4.95, seed=1)
    idx_s, idx_t = g["edge_index"]
    assert g["z"].numel() == 192 and g["lattice"].shape == (3, 3, 3)
    assert bool((idx_s[1:] >= idx_s[:-1]).all()), "edges grouped by centre"
    vec = g["pos"][idx_t] - g["pos"][idx_s] + torch.einsum("ni,nij->nj", g["edge_shift"], g["lattice"][g["batch"][idx_s]])
    d = vec.norm(dim=1)
    assert float(d.max()) <= 4.95 + 1e-5 and float(d.min()) > 0.5
    deg = torch.bincount(idx_s, minlength=192).float()
    assert 14 < float(deg.mean()) < 22
    # symmetric neighbour list
    key = set(zip(idx_s.tolist(), idx_t.tolist(), map(tuple, g["edge_shift"].int().tolist())))
    assert all((t, s, (-a, -b, -c)) in key for (s, t, (a, b, c)) in list(key)[:200])
    g2 = random_periodic_cells(2, 10, 6.0, 4.95, seed=3, max_neighbors=8)
    assert int(torch.bincount(g2["edge_index"][0]).max()) <= 8
    g3 = random_molecules(4, 9, 4.0, 6.0)
    assert float(g3["edge_shift"].abs().max()) == 0.0


def test_fully_linked_fallback_per_graph():
    """data/convert.py:154-157 + data/utils.py:10-20: a structure without any neighbour becomes fully linked; the
    helper applies it per graph of a batch and keeps edges grouped by centre."""
    from lcaonet_amd.data.neighbors import _link_isolated_graphs
    pos = torch.arange(21, dtype=torch.float32).view(7, 3)
    b = torch.tensor([0, 0, 1, 1, 1, 2, 3])                       # graph 1: 3 atoms, no edges; graph 2/3: single atoms
    e_src, e_dst = torch.tensor([0, 1]), torch.tensor([1, 0])
    out = _link_isolated_graphs(pos, b, 4, e_src, e_dst, torch.ones(2, 3), torch.tensor([5.0, 5.0]))
    s, t, sh, d = out
    assert list(zip(s.tolist(), t.tolist())) == [(0, 1), (1, 0), (2, 3), (2, 4), (3, 2), (3, 4), (4, 2), (4, 3)]
    assert sh[:2].eq(1).all() and sh[2:].eq(0).all()
    assert torch.allclose(d[2:], (pos[t[2:]] - pos[s[2:]]).norm(dim=1))
    same = _link_isolated_graphs(pos[:2], b[:2], 1, e_src, e_dst, torch.ones(2, 3), torch.tensor([5.0, 5.0]))
    assert same[0] is e_src
