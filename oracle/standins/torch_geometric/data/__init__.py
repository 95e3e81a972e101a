"""Attribute/dict graph containers: just enough of the PyG Data/Batch surface
that the reference's forward() touches (graph[key], graph[key]=v, graph.get)."""
import torch


class Data:
    def __init__(self, **kwargs):
        object.__setattr__(self, "_store", {})
        for k, v in kwargs.items():
            self._store[k] = v

    def __getattr__(self, key):
        store = object.__getattribute__(self, "_store")
        if key in store:
            return store[key]
        raise AttributeError(key)

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
            out._store[k] = v.to(*args, **kwargs) if torch.is_tensor(v) else v
        return out


class Batch(Data):
    @classmethod
    def from_data_list(cls, data_list):
        """Concatenate graphs; keys containing 'index' are offset by the node count."""
        out = cls()
        keys = data_list[0].keys()
        n_off = 0
        cols = {k: [] for k in keys}
        batch = []
        for i, d in enumerate(data_list):
            n = d["z"].size(0)
            for k in keys:
                v = d[k]
                if "index" in k:
                    v = v + n_off
                cols[k].append(v)
            batch.append(torch.full((n,), i, dtype=torch.long))
            n_off += n
        for k in keys:
            dim = -1 if "index" in k else 0
            out[k] = torch.cat(cols[k], dim=dim)
        out["batch"] = torch.cat(batch)
        return out


class Dataset(torch.utils.data.Dataset):
    def __init__(self, *a, **k):
        super().__init__()
