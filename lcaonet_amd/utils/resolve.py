"""String / type -> object resolvers: same names, normalisation and errors as the reference's
lcaonet/utils/resolve.py:16-147 (the only plugin mechanism the reference has: a user may hand
`LCAONet` a BaseRadialBasis / BaseCutoff subclass instead of a name)."""
from __future__ import annotations

from collections.abc import Callable
from inspect import getmembers, isfunction
from typing import Any

import torch

from .inits import glorot, glorot_orthogonal


def _normalize_string(s: str) -> str:
    return "".join(ch for ch in s.lower() if ch not in "-_ ")


def _resolver(query: Any, classes: list, base_cls: type | None = None, return_initialize: bool = True, **kwargs):
    """Look `query` up among `classes`: a (normalised) name, one of the classes / any subclass of `base_cls`, or — for
    plain callables such as init functions — the callable itself.  Classes are instantiated with **kwargs unless
    return_initialize is False.  ValueError for anything else, with the reference's two messages."""
    make = (lambda c: c(**kwargs)) if return_initialize else (lambda c: c)
    if isinstance(query, str):
        by_name = {_normalize_string(c.__name__): c for c in reversed(classes)}   # first match wins, as a linear scan
        if query in by_name:
            return make(by_name[query])
    elif isinstance(query, type):
        if any(query is c for c in classes) or (base_cls is not None and issubclass(query, base_cls)):
            return make(query)
    elif callable(query):
        if any(query is c for c in classes):
            return query
    else:
        raise ValueError(f"{query} must be str or type or class")
    raise ValueError(f"{query} not found")


def _activation_classes() -> list:
    from ..nn.activation import ShiftedSoftplus, Swish

    acts = [a for a in vars(torch.nn.modules.activation).values()
            if isinstance(a, type) and issubclass(a, torch.nn.Module)]
    return acts + [Swish, ShiftedSoftplus]


def activation_resolver(query: torch.nn.Module | str = "relu", **kwargs) -> torch.nn.Module:
    if isinstance(query, str):
        query = _normalize_string(query)
    return _resolver(query, _activation_classes(), torch.nn.Module, **kwargs)


# activation name -> nonlinearity name torch.nn.init.calculate_gain understands (resolve.py:88-98 of the reference:
# Swish borrows the sigmoid gain, ShiftedSoftplus and anything unknown the linear one)
_GAIN_OF = {"sigmoid": "sigmoid", "tanh": "tanh", "relu": "relu", "selu": "selu", "leakyrelu": "leaky_relu",
            "swish": "sigmoid", "shiftedsoftplus": "linear"}


def activation_gain_resolver(query: torch.nn.Module | str = "relu", **kwargs) -> float:
    """Initialisation gain for an activation given by name or class (reference resolve.py:79-105)."""
    if isinstance(query, str):
        query = _normalize_string(query)
    try:
        name = _resolver(query, _activation_classes(), torch.nn.Module, False).__name__.lower()
    except ValueError:
        name = ""
    return torch.nn.init.calculate_gain(_GAIN_OF.get(name, "linear"), **kwargs)


def init_resolver(query: Callable | str = "orthogonal") -> Callable:
    if isinstance(query, str):
        query = _normalize_string(query)
    funcs = [f[1] for f in getmembers(torch.nn.init, isfunction)] + [glorot, glorot_orthogonal]
    funcs = [f for f in funcs if "deprecated" not in str(f)]
    return _resolver(query, funcs, return_initialize=False)


def init_param_resolver(query: Callable) -> tuple[str, ...]:
    return query.__code__.co_varnames[: query.__code__.co_argcount]


def cutoffnet_resolver(query="polynomialcutoff", **kwargs):
    from ..nn import cutoff as C

    if isinstance(query, str):
        query = _normalize_string(query)
        if query[-6:] != "cutoff":
            query += "cutoff"
    cns = [c for c in vars(C).values() if isinstance(c, type) and issubclass(c, C.BaseCutoff)]
    return _resolver(query, cns, C.BaseCutoff, True, **kwargs)


def rbf_resolver(query="hydrogenradialbasis", **kwargs):
    from ..nn import rbf as R

    if isinstance(query, str):
        query = _normalize_string(query)
        if query[-11:] != "radialbasis":
            query += "radialbasis"
    rbfs = [c for c in vars(R).values() if isinstance(c, type) and issubclass(c, R.BaseRadialBasis)]
    return _resolver(query, rbfs, R.BaseRadialBasis, True, **kwargs)
