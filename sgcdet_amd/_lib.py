"""Loader of the gfx950 shared library.  Fails loudly: there is NO CPU fallback.

The library is built in-tree by ``sgcdet_amd.build`` (``__graft_entry__.build()``);
if it is missing and hipcc is available it is built on first use, otherwise the import
error says what to run.
"""
import os

# torch first: it ships its own libamdhip64; loading our library before torch would bind a
# second HIP runtime (/opt/rocm) into the process and the GPU would not be visible to it.
import torch  # noqa: F401

from ._abi import Library
from . import build as _build

_LIB = None


def library():
    global _LIB
    if _LIB is None:
        path = _build.LIB
        if not os.path.exists(path):
            try:
                _build.build()
            except Exception as e:  # pragma: no cover - depends on the toolchain
                raise ImportError(
                    f"sgcdet_amd: {path} is missing and could not be built ({e}). "
                    "Run `python -m sgcdet_amd.build` (needs hipcc, targets gfx950). "
                    "There is no CPU fallback for the product path.") from e
        _LIB = Library(path)
        if _LIB.backend != "hip-gfx950":
            raise ImportError(f"sgcdet_amd: {path} reports backend '{_LIB.backend}', expected 'hip-gfx950'")
    return _LIB
