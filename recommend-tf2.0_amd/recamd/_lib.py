"""Loader of the in-tree native build (librecamd.so + the pybind11 shim `_C`)."""
import importlib
import os

_here = os.path.dirname(os.path.abspath(__file__))
lib_path = os.path.join(_here, "librecamd.so")

if not os.path.exists(lib_path):
    raise ImportError(
        f"{lib_path} not found: the HIP extension is not built. Run `python -c 'import "
        "__graft_entry__ as g; g.build()'` (or `make -C recommend-tf2.0_amd/csrc`). "
        "There is no CPU fallback for the product path.")
try:
    C = importlib.import_module(__package__ + "._C")
except ImportError as e:  # pragma: no cover
    raise ImportError(f"recamd pybind11 shim failed to load ({e}); rebuild with __graft_entry__.build()") from e
shim_path = C.__file__
