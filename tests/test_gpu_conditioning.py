"""Conditioning of the per-triplet normalisation outside the random-init regime.

lcaonet.py:179-184 normalises every per-triplet vector v = sum_l Y_l(cos theta) A_l directly (F.normalize).  The fused
three-body kernel never forms v per triplet: it takes |v| from a per-edge 10-float record.  With random-init weights the
per-l vectors A_l are nearly orthogonal and any form works; after training they can be strongly correlated, and then
v passes close to zero for some angles.  This test drives that regime: f_coeffs.2.weight = rank 1 + 1e-3 noise makes the
A_l of an edge nearly collinear, and a graph with ~10^5 triplets sweeps the angle through the cancellation.
Compared, all against the fp64 oracle on the same weights:
  * the fused kernel (lcao_triplet_gemm),
  * the literal per-triplet norm (lcao_triplet_sum, exact_norm=True),
  * the fp32 oracle itself (the reference's own arithmetic)."""
import pytest
import torch

from oracle import restatement as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-300))


def _collinear_model(dev, width, seed, noise):
    from lcaonet_amd.model import LCAONet
    kw = dict(emb_size=64, emb_size_coeff=width, emb_size_conv=width, n_interaction=2, cutoff=5.0, max_z=36)
    torch.manual_seed(seed)
    m = LCAONet(**kw).eval()
    gen = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for mod in m.modules():
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
                mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
        for layer in m.int_layers:
            w = layer.f_coeffs[2].weight                 # (V, V): rank 1 + noise  ->  per-l vectors nearly collinear
            a, b = torch.randn(w.shape[0], 1, generator=gen), torch.randn(1, w.shape[1], generator=gen)
            w.copy_((a @ b) / w.shape[1] ** 0.5 + noise * torch.randn(w.shape, generator=gen) / w.shape[1] ** 0.5)
    m.invalidate_caches()
    sd = {k: v.clone().detach() for k, v in m.state_dict().items()}
    return m.to(dev).requires_grad_(False), sd, kw


@pytest.mark.parametrize("width", [128, 64])
@pytest.mark.parametrize("noise", [1e-3, 1e-1])
def test_three_body_norm_with_correlated_orbital_vectors(dev, width, noise):
    from lcaonet_amd.data.synthetic import random_periodic_cells
    m, sd, kw = _collinear_model(dev, width, 900 + width, noise)
    g = random_periodic_cells(6, 20, 6.6, 0.99 * 5.0, seed=31)
    cfg = O.OracleConfig(**{**kw, "rbf_type": "hydrogen"})
    tr64, tr32 = {}, {}
    with torch.no_grad():
        O.forward(sd, cfg, g.to_dict(), dtype=torch.float64, trace=tr64)
        O.forward(sd, cfg, g.to_dict(), dtype=torch.float32, trace=tr32)
    trk, trl = {}, {}
    with torch.no_grad():
        m(g.to(dev), trace=trk)                      # fused kernel
        m(g.to(dev), trace=trl, exact_norm=True)     # literal per-triplet norm
    order = trk["order"].long().cpu()
    worst = 0.0
    for i in (1, 2):
        want = tr64[f"three_body_w{i}"][order]
        e_kernel = _relerr(trk[f"three_body_w{i}"], want)
        e_literal = _relerr(trl[f"three_body_w{i}"], want)
        e_ref32 = _relerr(tr32[f"three_body_w{i}"][order], want)
        print(f"V={width} noise={noise:g} layer {i}: fused {e_kernel:.2e}  literal {e_literal:.2e}  fp32 oracle {e_ref32:.2e}")
        # the fused form may not lose more than 4x against the better-conditioned forms (and never more than 1e-5 absolute
        # headroom when all of them sit at rounding level)
        assert e_kernel <= max(4.0 * max(e_literal, e_ref32), 1e-5)
        worst = max(worst, e_kernel)
    assert worst < 1e-2
