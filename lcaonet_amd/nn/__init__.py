from .activation import ShiftedSoftplus, Swish
from .base import Dense

__all__ = ["Swish", "ShiftedSoftplus", "Dense"]
