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


LONG_SEGMENT = 512   # average rows per segment above which the reduction is done in two levels


def _segment_sum_two_level(rows: Tensor, ptr: Tensor, dim_size: int, chunk: int = 256) -> Tensor:
    """Few, long segments (e.g. per-graph sums over 10^5 atoms): split every segment into chunks of <= `chunk` rows,
    sum the chunks in parallel (one wave each), then sum each segment's partials.  Fixed order => deterministic."""
    p64 = ptr.long()
    seg_len = p64[1:] - p64[:-1]
    n_chunks = (seg_len + chunk - 1) // chunk
    cptr = torch.zeros(dim_size + 1, dtype=torch.long, device=rows.device)
    cptr[1:] = torch.cumsum(n_chunks, 0)
    total = (rows.shape[0] + chunk - 1) // chunk + dim_size          # upper bound on the number of chunks, no sync
    cid = torch.arange(total + 1, device=rows.device)
    seg_of = torch.searchsorted(cptr[1:], cid, right=True).clamp(max=max(dim_size - 1, 0))
    start = p64[seg_of] + (cid - cptr[seg_of]) * chunk
    lo = torch.minimum(start, p64[seg_of + 1])
    lo = torch.where(cid < cptr[-1], lo, p64[-1].expand_as(lo))     # chunks past the real count are empty
    # chunks are consecutive in memory, so one pointer array suffices: ptr2[c] = lo[c], ptr2[last] = end of rows
    ptr2 = torch.cat([lo[:total], p64[-1:]]).to(torch.int32)
    partial = ops.segment_sum(rows, ptr2)                             # (total, width)
    return ops.segment_sum(partial, torch.clamp(cptr, max=total).to(torch.int32))


def index_plan(index: Tensor, dim_size: int):
    """(order, ptr): the stable argsort of `index` (int32) and the segment bounds of the sorted list — what `scatter` builds
    from an index on every call; callers that reduce over the same index repeatedly (the per-graph readout over the batch
    index) build it once and pass it as `plan`."""
    order = torch.argsort(index, stable=True)
    return order.to(torch.int32), csr_ptr(index[order], dim_size)


def scatter(src: Tensor, index: Tensor, dim: int = 0, dim_size: int | None = None, reduce: str = "sum", plan=None) -> Tensor:
    if dim != 0:
        raise NotImplementedError("scatter: the path only reduces along dim 0")
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0  # host sync, as torch_scatter does
    tail = src.shape[1:]
    src2 = src.reshape(src.shape[0], -1) if src.dim() != 2 else src
    order, ptr = plan if plan is not None else index_plan(index, dim_size)
    rows = ops.gather_rows(src2.float().contiguous(), order)
    if dim_size > 0 and rows.shape[0] > LONG_SEGMENT * dim_size:
        out = _segment_sum_two_level(rows, ptr, dim_size)
    else:
        out = ops.segment_sum(rows, ptr)
    if reduce in ("sum", "add"):
        pass
    elif reduce == "mean":
        cnt = (ptr[1:] - ptr[:-1]).clamp(min=1).to(out.dtype)
        out = out / cnt.unsqueeze(1)
    else:
        raise ValueError(f"reduce={reduce} not supported")
    return out.reshape((dim_size,) + tuple(tail))
