import json
import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def dev():
    """cuda:0 with the HIP library loaded (-m gpu tests)."""
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    from lcaonet_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def relerr(a, b) -> float:
    """max|a-b| / max|b|  — the per-tensor parity metric (SURVEY.md §7 hard part 2)."""
    a = torch.as_tensor(a).double().cpu()
    b = torch.as_tensor(b).double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    if b.numel() == 0:
        return 0.0
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


class Golden:
    """One tests/golden/<name>.npz produced by oracle/make_golden.py from the reference."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.cfg = json.loads(str(self.z["cfg"])) if "cfg" in self.z.files else None

    def graph(self, dtype=torch.float32):
        g = {}
        for k in self.z.files:
            if k.startswith("in/"):
                v = torch.from_numpy(self.z[k])
                g[k[3:]] = v.to(dtype) if v.is_floating_point() else v
        return g

    def state_dict(self):
        sd = {k[3:]: torch.from_numpy(self.z[k]) for k in self.z.files if k.startswith("sd/")}
        for key in ("atomref", "mean"):
            if self.cfg.get(key) is not None:
                sd["pp_layer." + key] = torch.tensor(self.cfg[key], dtype=torch.float32)
        return sd

    def ref(self, key, tag="f32"):
        return torch.from_numpy(self.z[f"{tag}/{key}"])

    def has(self, key, tag="f32"):
        return f"{tag}/{key}" in self.z.files


MODEL_CASES = [
    "fixture_default", "fixture_valence", "fixture_valence_extend", "fixture_n_per_orb2", "fixture_max_orb4p",
    "fixture_bessel", "fixture_cosine_noelec_mean", "fixture_atomref", "fixture_dims", "fixture_direct_forces",
    "fixture_autograd_forces", "cells_h32", "cells_h16_train", "cells_z84_valence",
    "fixture_shiftedsoftplus", "cells_h16_tanh_autograd_forces",       # round 2: activations other than SiLU
]


@pytest.fixture(scope="session")
def golden_loader():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return load
