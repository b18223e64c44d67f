"""tdoa_amd -- host-side mirror of the reference's correlation call surface over the
MI355X C-ABI library (include/tdoa_mi355x.h).  PyTorch is used by bench.py only for
device buffers and torch.distributed; this package needs numpy + the HIP library."""
from . import build, capi, sharding  # noqa: F401
from .capi import Context, TdoaError  # noqa: F401

__all__ = ["build", "capi", "sharding", "Context", "TdoaError"]
