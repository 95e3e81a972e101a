"""scatter(src, index, dim=0, dim_size, reduce) with torch_scatter's semantics for the call sites on the
path (lcaonet.py:189,208-213,293,307; post.py:65,71,83): rows are ordered by a stable sort on `index` and
reduced per contiguous segment by the HIP kernel — deterministic, no float atomics."""
from __future__ import annotations

import torch
from torch import Tensor

from .. import ops


def csr_ptr(sorted_index: Tensor, n: int) -> Tensor:
    """int32 row pointer of a sorted index vector, without a host sync."""
    bounds = torch.arange(n + 1, device=sorted_index.device, dtype=sorted_index.dtype)
    return torch.searchsorted(sorted_index, bounds).to(torch.int32)


def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: int | None = None, reduce: str = "sum") -> Tensor:
    if dim != 0:
        raise NotImplementedError("scatter: the path only reduces along dim 0")
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0  # host sync, as torch_scatter does
    tail = src.shape[1:]
    src2 = src.reshape(src.shape[0], -1) if src.dim() != 2 else src
    order = torch.argsort(index, stable=True)
    sorted_idx = index[order]
    rows = ops.gather_rows(src2.float().contiguous(), order.to(torch.int32))
    ptr = csr_ptr(sorted_idx, dim_size)
    out = ops.segment_sum(rows, ptr)
    if reduce in ("sum", "add"):
        pass
    elif reduce == "mean":
        cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(out.dtype)
        out = out / cnt.unsqueeze(1)
    else:
        raise ValueError(f"reduce={reduce} not supported")
    return out.reshape((dim_size,) + tuple(tail))
