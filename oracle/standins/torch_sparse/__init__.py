"""Stand-in for the sliver of torch_sparse.SparseTensor that get_triplets uses."""
import torch


class _Storage:
    def __init__(self, row, col, value):
        self._row, self._col, self._value = row, col, value

    def row(self):
        return self._row

    def col(self):
        return self._col

    def value(self):
        return self._value


class SparseTensor:
    def __init__(self, row, col, value=None, sparse_sizes=None, _sorted=False):
        n_row, n_col = sparse_sizes
        if not _sorted:
            key = row * n_col + col
            perm = torch.argsort(key, stable=True)
            row, col = row[perm], col[perm]
            value = value[perm] if value is not None else None
        self._sizes = (n_row, n_col)
        self.storage = _Storage(row, col, value)
        self._rowptr = torch.zeros(n_row + 1, dtype=torch.long, device=row.device)
        self._rowptr[1:] = torch.cumsum(torch.bincount(row, minlength=n_row), 0)

    def __getitem__(self, idx):
        """Row-select: output row i is input row idx[i] (order and duplicates kept)."""
        start = self._rowptr[idx]
        cnt = self._rowptr[idx + 1] - start
        total = int(cnt.sum().item())
        new_row = torch.repeat_interleave(torch.arange(idx.numel(), device=idx.device), cnt)
        out_ptr = torch.cumsum(cnt, 0) - cnt
        offs = torch.arange(total, device=idx.device) - out_ptr[new_row]
        src = start[new_row] + offs
        col = self.storage._col[src]
        val = self.storage._value[src] if self.storage._value is not None else None
        return SparseTensor(new_row, col, val, (idx.numel(), self._sizes[1]), _sorted=True)

    def set_value(self, value, layout=None):
        return SparseTensor(self.storage._row, self.storage._col, value, self._sizes, _sorted=True)

    def sum(self, dim):
        assert dim == 1
        if self.storage._value is None:
            return (self._rowptr[1:] - self._rowptr[:-1]).to(torch.float32)
        out = torch.zeros(self._sizes[0], dtype=self.storage._value.dtype)
        return out.index_add_(0, self.storage._row, self.storage._value)
