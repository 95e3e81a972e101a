"""lcaonet_amd — MI355X-native implementation of LCAONet's per-edge interaction hot path.

Mirrors the reference package layout for the pieces on the path (`lcaonet_amd.model.LCAONet`,
`lcaonet_amd.nn.{rbf,shbf,cutoff,...}`, `lcaonet_amd.atomistic`, `lcaonet_amd.data.keys`).  All numerics run
in liblcaonet_hip.so (hand-written HIP for gfx950); importing the package does not load the library, the
first op does — and raises if it is missing (there is no CPU fallback)."""
__version__ = "0.1.0"
