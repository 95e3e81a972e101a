"""Minimal graph containers with the slice of the PyG `Data` / `Batch` surface that
`LCAONet.forward` touches: `graph[key]`, `graph[key] = v`, `graph.get(key)`, attribute access,
`.to(device)`.  A real `torch_geometric.data.Batch` works just as well (it is not installed in
this image, so the package carries its own)."""
from __future__ import annotations

import torch


class Data:
    def __init__(self, **kwargs):
        object.__setattr__(self, "_store", dict(kwargs))

    def __getattr__(self, key):
        store = object.__getattribute__(self, "_store")
        try:
            return store[key]
        except KeyError:
            raise AttributeError(key) from None

    def __setattr__(self, key, value):
        self._store[key] = value

    def __getitem__(self, key):
        return self._store[key]

    def __setitem__(self, key, value):
        self._store[key] = value

    def __contains__(self, key):
        return key in self._store

    def get(self, key, default=None):
        return self._store.get(key, default)

    def keys(self):
        return list(self._store.keys())

    def to(self, *args, **kwargs):
        out = self.__class__()
        for k, v in self._store.items():
            if k.startswith("_") and k.endswith("_memo"):      # tied to one batch tensor (model/base.py::num_graphs, batch_plan)
                continue
            out._store[k] = v.to(*args, **kwargs) if torch.is_tensor(v) else v
        return out

    def to_dict(self):
        return {k: v for k, v in self._store.items() if k != "num_graphs" and not (k.startswith("_") and k.endswith("_memo"))}

    def __repr__(self):
        items = ", ".join(f"{k}={tuple(v.shape) if torch.is_tensor(v) else v}" for k, v in self._store.items())
        return f"{self.__class__.__name__}({items})"


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        """Concatenate graphs the way PyG does: node-indexed keys (names containing 'index')
        are offset by the running node count and concatenated along the last dim."""
        out = cls()
        keys = [k for k in data_list[0].keys() if torch.is_tensor(data_list[0][k])]
        cols = {k: [] for k in keys}
        batch, n_off = [], 0
        for i, d in enumerate(data_list):
            n = d["z"].size(0)
            for k in keys:
                v = d[k]
                cols[k].append(v + n_off if "index" in k else v)
            batch.append(torch.full((n,), i, dtype=torch.long, device=d["z"].device))
            n_off += n
        for k in keys:
            out[k] = torch.cat(cols[k], dim=-1 if "index" in k else 0)
        out["batch"] = torch.cat(batch)
        out["num_graphs"] = len(data_list)
        return out
