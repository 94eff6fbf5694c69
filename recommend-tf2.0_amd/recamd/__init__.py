"""recamd — host side of the MI355X embedding-lookup + feature-interaction library.

``recamd.ops`` wraps the C ABI (include/recamd.h, librecamd.so) for torch tensors: torch is only
used for device memory and streams.  There is NO CPU fallback: importing this package without the
in-tree HIP build, or calling an op on a non-GPU tensor, raises.
"""
from ._lib import C, lib_path, shim_path  # noqa: F401  (raises ImportError loudly when not built)
from . import ops  # noqa: F401

__version__ = "0.1.0"
