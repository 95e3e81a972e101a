"""LCAONet.forward_sharded through REAL torch.distributed collectives on the GPU path: two / three ranks share cuda:0 and
talk over gloo (on a node each rank has its own GPU and the backend is nccl = RCCL; shard._a2a stages device tensors through
the host for gloo).  Exercises what the lock-step tests cannot: shard.start_exchange's asynchronous all-to-all with the next
layer's owned-edge coefficient kernel enqueued before the wait, reduce_graph_outputs, and the reverse exchange of the
sharded autograd-force sweep."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import relerr

pytestmark = pytest.mark.gpu
KW = dict(emb_size=64, emb_size_coeff=64, emb_size_conv=64, n_interaction=3, cutoff=4.0)


def _model(dev, **extra):
    from test_gpu_parity import _random_model
    return _random_model(dev, 2024, **dict(KW, **extra))[0]


def _graph(dev):
    from lcaonet_amd.data.synthetic import random_periodic_cells
    return random_periodic_cells(1, 60, 10.0, 0.99 * 4.0, seed=9).to(dev)


def _worker(rank, world, port, mode, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from lcaonet_amd import shard
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        extra = {"energy": {}, "direct": dict(regress_forces=True, direct_forces=True),
                 "autograd": dict(regress_forces=True, direct_forces=False)}[mode]
        m = _model(dev, **extra)
        plan = shard.make_plans(_graph(dev), world)[rank]
        with torch.no_grad():
            out = m.forward_sharded(plan)
        torch.cuda.synchronize()
        if mode == "energy":
            ret[rank] = (out.cpu(), None, plan.owned_global.cpu(), plan.n_ghost)
        else:
            ret[rank] = (out[0].cpu(), out[1].cpu(), plan.owned_global.cpu(), plan.n_ghost)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "energy"), (3, "energy"), (2, "direct"), (2, "autograd")])
def test_forward_sharded_over_process_group_matches_unsharded(dev, world, mode):
    extra = {"energy": {}, "direct": dict(regress_forces=True, direct_forces=True),
             "autograd": dict(regress_forces=True, direct_forces=False)}[mode]
    m = _model(dev, **extra)
    with torch.no_grad():
        want = m(_graph(dev))
    want_e, want_f = (want, None) if mode == "energy" else want
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, mode, ret), nprocs=world, join=True)
    assert len(ret) == world
    seen = torch.zeros(want_e.new_zeros(1).numel() and 60, dtype=torch.bool)
    for r in range(world):
        e, f, owned, n_ghost = ret[r]
        assert n_ghost > 0
        assert relerr(e, want_e) <= 5e-6, (r, relerr(e, want_e))
        if f is not None:
            assert relerr(f, want_f[owned.to(want_f.device)]) <= (2e-5 if mode == "autograd" else 5e-6), r
        seen[owned] = True
    assert bool(seen.all())
