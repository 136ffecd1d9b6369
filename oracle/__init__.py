"""CPU oracle package -- TEST INFRASTRUCTURE ONLY.

Importers allowed: ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py``.  The product package ``sgcdet_amd`` never imports this.

``oracle.ops()`` returns a ``TensorOps`` front end (CPU tensors) over
``libsgc_oracle.so`` (scalar checker) or ``libsgc_oracle_omp.so`` (OpenMP build used
for the CPU baseline timing).  Parity status: see the header of ``sgc_oracle.c``
("parity unpinned" by the reference's own tests; pinned by the grid_sample identity
and by container-generated golden vectors).
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_CACHE = {}


def build(force=False):
    """Compile the C restatement with gcc (seconds)."""
    targets = [os.path.join(_HERE, n) for n in ("libsgc_oracle.so", "libsgc_oracle_omp.so")]
    deps = [os.path.join(_HERE, "sgc_oracle.c"), os.path.join(_HERE, "..", "include", "sgcdet_amd.h")]
    newest = max(os.path.getmtime(d) for d in deps if os.path.exists(d))
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < newest for t in targets)
    if stale:
        subprocess.run(["make", "-C", _HERE, "-s", "-B"], check=True)
    return targets


def build_ref():
    """oracle/_ref: the reference's rotated-box IoU header compiled with g++ from /root/reference (build container
    only).  Returns the path of the shared object, or None where the reference tree is absent and no prebuilt
    copy travelled with the snapshot."""
    so = os.path.join(_HERE, "_ref", "libref_box_iou.so")
    if os.path.isdir("/root/reference"):
        subprocess.run(["make", "-C", _HERE, "-s", "_ref"], check=True)
    return so if os.path.exists(so) else None


def ref_box_iou_rotated(a, b, f64=False, variant="cuda"):
    """IoU matrix of rotated boxes (xc, yc, w, h, radians) by the REFERENCE's own code (oracle/_ref); None if absent.
    ``variant``: which branch of the header's convex-hull sort -- "cuda" (the exchange sort of the kernels the reference
    runs) or "cpu" (std::sort)."""
    import ctypes
    import numpy as np
    so = build_ref()
    if so is None:
        return None
    dll = ctypes.CDLL(so)
    dt = np.float64 if f64 else np.float32
    a = np.ascontiguousarray(a, dtype=dt)
    b = np.ascontiguousarray(b, dtype=dt)
    out = np.empty((a.shape[0], b.shape[0]), dtype=dt)
    fn = getattr(dll, ("ref_box_iou_rotated_f64_" if f64 else "ref_box_iou_rotated_") + variant)
    fn.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 2
    fn.restype = None
    fn(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.shape[0], b.shape[0])
    return out


def library(omp=False):
    key = "omp" if omp else "scalar"
    if key not in _CACHE:
        from sgcdet_amd._abi import Library
        path = os.path.join(_HERE, "libsgc_oracle_omp.so" if omp else "libsgc_oracle.so")
        if not os.path.exists(path):
            build()
        _CACHE[key] = Library(path)
    return _CACHE[key]


def ops(omp=False):
    from sgcdet_amd.tensor_api import TensorOps
    return TensorOps(library(omp), "cpu")
