from .capture import CapturedForward
from .lcaonet import LCAONet

__all__ = ["LCAONet", "CapturedForward"]
