from .lcaonet import LCAONet

__all__ = ["LCAONet"]
