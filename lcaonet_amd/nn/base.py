"""Dense = nn.Linear with a pluggable initialiser (reference: lcaonet/nn/base.py:11-81).  Same parameter
names (`weight`, `bias`), so reference checkpoints load unchanged.  forward() runs the HIP dense kernel."""
from __future__ import annotations

from collections.abc import Callable

import torch
import torch.nn as nn
from torch import Tensor

from .. import ops
from ..utils.resolve import init_param_resolver


def act_code(activation: nn.Module | None) -> int | None:
    """Kernel enum of an activation module, or None when it must run as a separate elementwise op."""
    if activation is None:
        return ops.ACT_NONE
    if type(activation) is nn.SiLU:
        return ops.ACT_SILU
    if type(activation) is nn.Sigmoid:
        return ops.ACT_SIGMOID
    return None


# Weight-derived caches kept in a module's __dict__ (never in state_dict): LCAOEmbedding's species / pair tables and
# LCAOInteraction's folded node weights.  They are keyed on (data_ptr, _version) of the tensors they were built from,
# which an in-place write THROUGH `.data` does not change — so everything that rewrites weights wholesale drops them.
WEIGHT_CACHE_KEYS = ("_species", "_fold", "_wt")


def drop_weight_caches(module) -> None:
    """Forget every weight-derived cache below `module` (they are rebuilt by the next forward)."""
    for m in module.modules():
        for k in WEIGHT_CACHE_KEYS:
            m.__dict__.pop(k, None)


class Dense(nn.Linear):
    def __init__(self, in_dim: int, out_dim: int, bias: bool = True,
                 weight_init: Callable[[Tensor], Tensor] | None = None,
                 bias_init: Callable[[Tensor], Tensor] | None = nn.init.zeros_, **kwargs):
        if bias and bias_init is None:
            raise ValueError("bias_init must not be None if set bias")
        self.bias_init = bias_init
        self.weight_init = weight_init
        if weight_init is not None:
            for p in init_param_resolver(weight_init):
                if p in kwargs:
                    continue
                if p == "gain":
                    kwargs[p] = 1.0
                elif p == "scale":  # glorot_orthogonal
                    kwargs[p] = 2.0
        self.kwargs = kwargs
        self._ready = False
        super().__init__(in_dim, out_dim, bias)
        self._ready = True
        self.reset_parameters()

    def reset_parameters(self):
        if not getattr(self, "_ready", False):
            # nn.Linear.__init__ calls this before our attributes exist; keep torch's default init so that
            # weight_init=None never leaves uninitialised memory (a quirk of the reference we do not copy)
            super().reset_parameters()
            return
        if self.weight_init is not None:
            self.weight_init(self.weight, **self.kwargs)
        if self.bias is not None and self.bias_init is not None:
            self.bias_init(self.bias)

    def extra_repr(self) -> str:
        return "in_features={}, out_features={}, bias={}, weight_init={}".format(
            self.in_features, self.out_features, self.bias is not None,
            self.weight_init.__name__ if self.weight_init is not None else None)

    def forward(self, x: Tensor, act: int = ops.ACT_NONE, residual: Tensor | None = None) -> Tensor:
        """act(x W^T + b) + residual on the HIP dense kernel.  When autograd is on and the input or a parameter requires
        grad, the layer runs as a twice-differentiable autograd Function over the same kernel (model/double.py: backward =
        lcao_dense on W^T + lcao_wgrad), so composing Dense layers trains exactly as nn.Linear does."""
        if torch.is_grad_enabled() and (x.requires_grad or self.weight.requires_grad
                                        or (self.bias is not None and self.bias.requires_grad)):
            from ..model.double import dense as dense_fn

            y = dense_fn(x, self.weight, self.bias)
            if act == ops.ACT_SILU:
                y = torch.nn.functional.silu(y)
            elif act == ops.ACT_SIGMOID:
                y = torch.sigmoid(y)
            return y if residual is None else y + residual
        return ops.dense(x, self.weight.detach(), None if self.bias is None else self.bias.detach(), act, residual)


def inference_only(fn):
    """Decorator for module forwards that run the HIP inference kernels and record NO autograd graph (the operator-level
    LCAOInteraction.forward / LCAOOut.forward).  The reference's versions are differentiable; rather than hand back a
    tensor whose gradients silently miss these parameters, the call refuses when autograd is on and a parameter of the
    module requires grad.  Under torch.no_grad() or with frozen parameters it runs as usual."""
    import functools

    @functools.wraps(fn)
    def wrapper(self, *args, **kwargs):
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise RuntimeError(
                f"{type(self).__name__}.{fn.__name__} runs the HIP inference kernels and records no autograd graph: call "
                "it under torch.no_grad() (or freeze the parameters), or go through LCAONet.forward, which is "
                "differentiable whenever autograd is on")
        with torch.no_grad():
            return fn(self, *args, **kwargs)

    return wrapper


def run_mlp(seq: nn.Sequential, x: Tensor) -> Tensor:
    """Run an nn.Sequential of Dense / activation modules, fusing each activation the kernel knows into the
    preceding Dense launch."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, Dense):
            nxt = mods[i + 1] if i + 1 < len(mods) else None
            code = act_code(nxt) if (nxt is not None and not isinstance(nxt, Dense)) else None
            if code is not None and nxt is not None and not isinstance(nxt, Dense):
                x = m(x, act=code)
                i += 2
                continue
            x = m(x)
        else:
            x = m(x)
        i += 1
    return x
