"""Stand-in for torch_scatter.scatter (sum / mean along dim 0..)."""
import torch


def scatter(src, index, dim=0, out=None, dim_size=None, reduce="sum"):
    if dim < 0:
        dim += src.dim()
    if dim_size is None:
        dim_size = int(index.max().item()) + 1 if index.numel() > 0 else 0
    shape = list(src.shape)
    shape[dim] = dim_size
    res = torch.zeros(shape, dtype=src.dtype, device=src.device)
    res.index_add_(dim, index, src)
    if reduce in ("sum", "add"):
        return res
    if reduce == "mean":
        cnt = torch.zeros(dim_size, dtype=src.dtype, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=src.dtype))
        cnt = cnt.clamp(min=1)
        view = [1] * src.dim()
        view[dim] = dim_size
        return res / cnt.view(view)
    raise ValueError(reduce)
