"""Data-parallel training across ranks (torch DistributedDataParallel over the HIP training path).  Batches of independent
crystals shard by whole graphs (zero halo), so multi-GPU training is plain DDP: every rank runs
model/train.py on its own graphs and the parameter gradients are all-reduced.  One GPU is available to the tests: two
ranks share cuda:0 and talk over gloo (on a node the launcher gives one GPU per rank and the backend is nccl = RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
KW = dict(emb_size=32, emb_size_coeff=64, emb_size_conv=64, n_interaction=2, cutoff=4.0)


def _model(dev):
    from lcaonet_amd.model import LCAONet
    torch.manual_seed(5)
    m = LCAONet(**KW)
    gen = torch.Generator().manual_seed(6)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm1d):
            mod.running_mean.copy_(torch.randn(mod.running_mean.shape, generator=gen) * 0.1)
            mod.running_var.copy_(torch.rand(mod.running_var.shape, generator=gen) + 0.5)
    return m.to(dev).eval()          # eval-mode BatchNorm: the per-rank and the pooled statistics coincide


def _batch(rank, dev):
    from lcaonet_amd.data.synthetic import random_periodic_cells
    return random_periodic_cells(3, 10, 6.0, 0.99 * 4.0, seed=40 + rank).to(dev)


def _loss(model, g, differentiable=True):
    out = model(g, differentiable=differentiable)
    return ((out - 0.5) ** 2).mean()


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        model = _model(dev)
        ddp = torch.nn.parallel.DistributedDataParallel(model)
        loss = _loss(ddp, _batch(rank, dev))
        loss.backward()
        grads = {n: p.grad.detach().cpu().clone() for n, p in model.named_parameters() if p.grad is not None}
        opt = torch.optim.SGD(ddp.parameters(), lr=1e-2)
        opt.step()
        ret[rank] = (float(loss.detach()), grads, {n: p.detach().cpu().clone() for n, p in model.named_parameters()})
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_ddp_two_ranks_average_the_hip_gradients():
    assert torch.cuda.is_available()
    dev = torch.device("cuda", 0)
    # single-process reference: the gradient of each rank's loss, averaged by hand
    want = None
    for r in range(2):
        m = _model(dev)
        _loss(m, _batch(r, dev)).backward()
        g = {n: p.grad.detach().cpu() for n, p in m.named_parameters() if p.grad is not None}
        want = g if want is None else {n: (want[n] + g[n]) / 2 for n in want}
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(2, _free_port(), ret), nprocs=2, join=True)
    assert len(ret) == 2
    (l0, g0, p0), (l1, g1, p1) = ret[0], ret[1]
    assert l0 != l1                                   # different graphs on the two ranks
    checked = 0
    for n, w in want.items():
        assert torch.equal(g0[n], g1[n]), n           # all-reduced: identical on both ranks
        if float(w.abs().max()) < 1e-12:
            continue
        assert float((g0[n] - w).abs().max() / w.abs().max()) <= 1e-5, n
        checked += 1
    assert checked > 25
    for n in p0:
        assert torch.equal(p0[n], p1[n]), n           # replicas stay in sync after the optimiser step
