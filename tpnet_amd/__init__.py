"""tpnet_amd -- MI355X (gfx950) implementation of TPNet's temporal-walk-matrix hot path.

Only what the path needs: the HIP kernels + C ABI (csrc/, include/tpnet_hip.h), the ctypes binding (_lib) and the
drop-in `RandomProjectionModule` mirroring /root/reference/models/TPNet.py:9-157.
"""
from .random_projection import RandomProjectionModule  # noqa: F401
from ._lib import TPNetHipError, load as load_library  # noqa: F401

__all__ = ["RandomProjectionModule", "TPNetHipError", "load_library"]
