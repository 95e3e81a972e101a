"""Stand-in for torch_geometric (test tooling; see oracle/standins/README.md)."""
from . import data, nn  # noqa: F401
